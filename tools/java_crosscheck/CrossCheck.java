import gr.iti.mklab.visual.datastructures.AbstractSearchStructure;
import gr.iti.mklab.visual.datastructures.IVFPQ;
import gr.iti.mklab.visual.datastructures.Linear;
import gr.iti.mklab.visual.datastructures.PQ;
import gr.iti.mklab.visual.datastructures.PQ.TransformationType;
import gr.iti.mklab.visual.utilities.Answer;
import gr.iti.mklab.visual.utilities.RandomPermutation;
import gr.iti.mklab.visual.utilities.RandomRotation;

import java.io.BufferedReader;
import java.io.File;
import java.io.FileReader;
import java.io.PrintWriter;
import java.util.ArrayList;
import java.util.List;

/**
 * Pins the CPU oracle of this repository (oracle/) to the REFERENCE's own classes.
 *
 * The build image has no JDK and none of the reference's jars, so the oracle is pinned only by hand-derived known answers
 * and by fixtures the oracle itself generated ("parity unpinned").  Anyone with a JDK, the reference jar and its four
 * dependencies (LingPipe 4.0.1, BDB-JE 5.0.58, EJML 0.23, Trove 3.0.3) closes that gap with this harness: it feeds the
 * committed fixtures (tests/golden/*.npz, exported to CSV by export_fixtures.py) to the reference's IVFPQ / PQ / Linear
 * through their public API -- loadCoarseQuantizer / loadProductQuantizer / indexVector / computeNearestNeighbors -- and dumps
 * ids and the raw bits of the distances; compare.py then checks them against the fixtures' expected answers, bit for bit.
 * It also prints the facts behind assumptions A1 (LingPipe BoundedPriorityQueue tie order: the flagged ivfpq_ties fixture) and
 * the JDK permutation known answers.
 *
 * This file only CALLS the reference; it contains none of its code.
 *
 *   java -cp <reference.jar>:<deps>:. CrossCheck <fixture dir produced by export_fixtures.py> <out dir>
 */
public class CrossCheck {

	static double[][] readMatrix(File f) throws Exception {
		List<double[]> rows = new ArrayList<double[]>();
		BufferedReader in = new BufferedReader(new FileReader(f));
		String line;
		while ((line = in.readLine()) != null) {
			if (line.isEmpty())
				continue;
			String[] s = line.split(",");
			double[] r = new double[s.length];
			for (int i = 0; i < s.length; i++)
				r[i] = Double.longBitsToDouble(Long.parseUnsignedLong(s[i], 16)); // exact: doubles travel as hex bits
			rows.add(r);
		}
		in.close();
		return rows.toArray(new double[0][]);
	}

	static int[] readMeta(File f) throws Exception { // one line: D,C,m,ks,w,k,transform,n,nq
		BufferedReader in = new BufferedReader(new FileReader(f));
		String[] s = in.readLine().split(",");
		in.close();
		int[] v = new int[s.length];
		for (int i = 0; i < s.length; i++)
			v[i] = Integer.parseInt(s[i]);
		return v;
	}

	static void deleteRecursively(File f) {
		File[] kids = f.listFiles();
		if (kids != null)
			for (File k : kids)
				deleteRecursively(k);
		f.delete();
	}

	static void dump(AbstractSearchStructure ix, double[][] queries, int k, File out) throws Exception {
		PrintWriter pw = new PrintWriter(out);
		for (double[] q : queries) {
			Answer a = ix.computeNearestNeighbors(k, q);
			String[] ids = a.getIds();
			double[] d = a.getDistances();
			StringBuilder sb = new StringBuilder();
			for (int i = 0; i < ids.length; i++) {
				if (i > 0)
					sb.append(',');
				sb.append(ids[i]).append(':').append(Long.toHexString(Double.doubleToLongBits(d[i])));
			}
			pw.println(sb);
		}
		pw.close();
	}

	static void runCase(File dir, File outDir) throws Exception {
		String name = dir.getName();
		int[] m = readMeta(new File(dir, "meta.csv"));
		int D = m[0], C = m[1], nsub = m[2], ks = m[3], w = m[4], k = m[5], tr = m[6];
		double[][] base = readMatrix(new File(dir, "base.csv"));
		double[][] queries = readMatrix(new File(dir, "queries.csv"));
		File bdb = new File(outDir, "bdb_" + name);
		deleteRecursively(bdb);
		bdb.mkdirs();
		TransformationType t = TransformationType.values()[tr];
		AbstractSearchStructure ix;
		if (name.startsWith("ivfpq")) {
			IVFPQ x = new IVFPQ(D, base.length, false, bdb.getPath(), nsub, ks, t, C, 64);
			x.loadCoarseQuantizer(new File(dir, "coarse_plain.csv").getPath());
			x.loadProductQuantizer(new File(dir, "pq_plain.csv").getPath());
			x.setW(w);
			ix = x;
		} else if (name.startsWith("pq")) {
			PQ x = new PQ(D, base.length, false, bdb.getPath(), nsub, ks, t, 64);
			x.loadProductQuantizer(new File(dir, "pq_plain.csv").getPath());
			ix = x;
		} else {
			ix = new Linear(D, base.length, false, bdb.getPath());
		}
		for (int i = 0; i < base.length; i++)
			ix.indexVector(Integer.toString(i), base[i]);
		dump(ix, queries, k, new File(outDir, name + ".answers.csv"));
		ix.close();
		if (t == TransformationType.RandomRotation) {
			// the matrix is EJML's (RandomRotation.java:30-35; seed 1 as IVFPQ.java:136): rotating the unit vectors reveals its rows,
			// and compare.py hands them to the oracle
			RandomRotation rr = new RandomRotation(1, D);
			PrintWriter pw = new PrintWriter(new File(outDir, name + ".rotation.csv"));
			for (int i = 0; i < D; i++) {
				double[] e = new double[D];
				e[i] = 1.0;
				double[] row = rr.rotate(e);
				StringBuilder sb = new StringBuilder();
				for (int j = 0; j < D; j++) {
					if (j > 0)
						sb.append(',');
					sb.append(String.format("%016x", Double.doubleToLongBits(row[j])));
				}
				pw.println(sb);
			}
			pw.close();
		}
		System.out.println(name + ": " + base.length + " vectors indexed, " + queries.length + " queries answered");
	}

	public static void main(String[] args) throws Exception {
		File in = new File(args[0]), out = new File(args[1]);
		out.mkdirs();
		// KAT-3 (RandomPermutation.java:29-56): permuting the identity vector reveals the index array
		PrintWriter pw = new PrintWriter(new File(out, "permutation.csv"));
		for (int dim : new int[] { 3, 8, 128 }) {
			double[] v = new double[dim];
			for (int i = 0; i < dim; i++)
				v[i] = i;
			double[] p = new RandomPermutation(1, dim).permute(v);
			StringBuilder sb = new StringBuilder(Integer.toString(dim));
			for (double x : p)
				sb.append(',').append((int) x);
			pw.println(sb);
		}
		pw.close();
		File[] cases = in.listFiles();
		java.util.Arrays.sort(cases);
		for (File c : cases)
			if (c.isDirectory())
				runCase(c, out);
	}
}
