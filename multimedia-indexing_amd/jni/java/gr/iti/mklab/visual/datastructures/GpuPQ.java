package gr.iti.mklab.visual.datastructures;

import gr.iti.mklab.visual.datastructures.PQ.TransformationType;
import gr.iti.mklab.visual.utilities.Result;

import java.io.BufferedReader;
import java.io.File;
import java.io.FileReader;

import com.aliasi.util.BoundedPriorityQueue;
import com.sleepycat.bind.tuple.IntegerBinding;
import com.sleepycat.bind.tuple.TupleBinding;
import com.sleepycat.bind.tuple.TupleInput;
import com.sleepycat.bind.tuple.TupleOutput;
import com.sleepycat.je.Cursor;
import com.sleepycat.je.Database;
import com.sleepycat.je.DatabaseConfig;
import com.sleepycat.je.DatabaseEntry;
import com.sleepycat.je.LockMode;
import com.sleepycat.je.OperationStatus;

/** Drop-in for {@link PQ} (exhaustive ADC, PQ.java:290-322) over libmmidx_hip.so; see GpuIVFPQ. */
public class GpuPQ extends AbstractSearchStructure {

	private final long handle;
	private final int numSubVectors, numProductCentroids;
	private final Database iidToPqDB;

	public GpuPQ(int vectorLength, int maxNumVectors, boolean readOnly, String BDBEnvHome, int numSubVectors,
			int numProductCentroids, TransformationType transformation, boolean countSizeOnLoad, int loadCounter,
			boolean loadIndexInMemory, long cacheSize) throws Exception {
		super(vectorLength, maxNumVectors, readOnly, countSizeOnLoad, loadCounter, loadIndexInMemory, cacheSize);
		if (vectorLength % numSubVectors > 0) { // PQ.java:148-150
			throw new Exception("The given number of subvectors is not valid!");
		}
		this.numSubVectors = numSubVectors;
		this.numProductCentroids = numProductCentroids;
		double[] rotation = null;
		if (transformation == TransformationType.RandomRotation) {
			rotation = org.ejml.ops.RandomMatrices.createOrthogonal(vectorLength, vectorLength,
					new java.util.Random(1)).getData(); // RandomRotation.java:30-35, seed PQ.java:108
		}
		handle = MmidxNative.create(MmidxNative.KIND_PQ, vectorLength, numSubVectors, numProductCentroids, 0,
				transformation.ordinal(), null, rotation, Integer.getInteger("mmidx.device", 0));
		createOrOpenBDBEnvAndDbs(BDBEnvHome);
		DatabaseConfig dbConf = new DatabaseConfig();
		dbConf.setReadOnly(readOnly);
		dbConf.setTransactional(transactional);
		dbConf.setAllowCreate(true);
		iidToPqDB = dbEnv.openDatabase(null, "adc", dbConf); // PQ.java:168
		if (loadIndexInMemory) {
			loadIndexInMemory();
		}
	}

	public GpuPQ(int vectorLength, int maxNumVectors, boolean readOnly, String BDBEnvHome, int numSubVectors,
			int numProductCentroids, TransformationType transformation, long cacheSize) throws Exception {
		this(vectorLength, maxNumVectors, readOnly, BDBEnvHome, numSubVectors, numProductCentroids, transformation,
				true, 0, true, cacheSize);
	}

	public void loadProductQuantizer(String filename) throws Exception { // PQ.java:210-223
		int dsub = vectorLength / numSubVectors;
		double[] flat = new double[numSubVectors * numProductCentroids * dsub];
		BufferedReader in = new BufferedReader(new FileReader(new File(filename)));
		for (int i = 0; i < numSubVectors * numProductCentroids; i++) {
			String[] s = in.readLine().split(",");
			for (int k = 0; k < dsub; k++)
				flat[i * dsub + k] = Double.parseDouble(s[k]);
		}
		in.close();
		MmidxNative.setPq(handle, flat);
	}

	protected void indexVectorInternal(double[] vector) throws Exception { // PQ.java:232-268
		if (vector.length != vectorLength) {
			throw new Exception("The dimensionality of the vector is wrong!");
		}
		int[] cell = new int[1];
		TupleOutput output = new TupleOutput(); // appendPersistentIndex PQ.java:491-502 / :510-521
		if (numProductCentroids <= 256) {
			byte[] code = new byte[numSubVectors];
			MmidxNative.addVector(handle, loadCounter, vector, cell, code);
			for (int i = 0; i < numSubVectors; i++)
				output.writeByte(code[i]);
		} else {
			short[] code = new short[numSubVectors];
			MmidxNative.addVectorShort(handle, loadCounter, vector, cell, code);
			for (int i = 0; i < numSubVectors; i++)
				output.writeShort(code[i]);
		}
		DatabaseEntry data = new DatabaseEntry();
		TupleBinding.outputToEntry(output, data);
		DatabaseEntry key = new DatabaseEntry();
		IntegerBinding.intToEntry(loadCounter, key);
		iidToPqDB.put(null, key, data);
	}

	protected BoundedPriorityQueue<Result> computeNearestNeighborsInternal(int k, double[] query) throws Exception {
		int[] iids = new int[k];
		double[] dists = new double[k];
		int[] count = new int[1];
		MmidxNative.search(handle, k, 1, query, iids, dists, count);
		BoundedPriorityQueue<Result> nn = new BoundedPriorityQueue<Result>(new Result(), k);
		for (int i = count[0] - 1; i >= 0; i--)
			nn.offer(new Result(iids[i], dists[i]));
		return nn;
	}

	protected BoundedPriorityQueue<Result> computeNearestNeighborsInternal(int k, int iid) throws Exception {
		// PQ.computeKnnSDC, PQ.java:334-374 (byte codes only: the reference dereferences pqByteCodes at :350)
		if (numProductCentroids > 256) {
			throw new Exception("Symmetric search needs byte codes (numProductCentroids <= 256)");
		}
		int[] iids = new int[k];
		double[] dists = new double[k];
		int[] count = new int[1];
		MmidxNative.searchSdc(handle, k, 1, new int[] { iid }, iids, dists, count);
		BoundedPriorityQueue<Result> nn = new BoundedPriorityQueue<Result>(new Result(), k);
		for (int i = count[0] - 1; i >= 0; i--)
			nn.offer(new Result(iids[i], dists[i]));
		return nn;
	}

	private void loadIndexInMemory() throws Exception { // PQ.java:436-483
		final int B = 1 << 16;
		final boolean bytes = numProductCentroids <= 256;
		int[] iids = new int[B];
		byte[] bcodes = bytes ? new byte[B * numSubVectors] : null;
		short[] scodes = bytes ? null : new short[B * numSubVectors];
		int n = 0, counter = 0;
		DatabaseEntry key = new DatabaseEntry(), data = new DatabaseEntry();
		Cursor cursor = iidToPqDB.openCursor(null, null);
		while (cursor.getNext(key, data, LockMode.DEFAULT) == OperationStatus.SUCCESS && counter < maxNumVectors) {
			TupleInput input = TupleBinding.entryToInput(data);
			iids[n] = counter++; // iid == position, PQ.java:303,318
			for (int i = 0; i < numSubVectors; i++) {
				if (bytes)
					bcodes[n * numSubVectors + i] = input.readByte();
				else
					scodes[n * numSubVectors + i] = input.readShort();
			}
			if (++n == B) {
				if (bytes)
					MmidxNative.addCodes(handle, n, iids, null, bcodes);
				else
					MmidxNative.addCodesShort(handle, n, iids, null, scodes);
				n = 0;
			}
		}
		cursor.close();
		if (n > 0) {
			if (bytes)
				MmidxNative.addCodes(handle, n, iids, null, bcodes);
			else
				MmidxNative.addCodesShort(handle, n, iids, null, scodes);
		}
	}

	@Override
	public void outputIndexingTimesInternal() {
	}

	@Override
	public void closeInternal() {
		iidToPqDB.close();
		MmidxNative.destroy(handle);
	}

	/** native flat snapshot (mmidx_save / mmidx_load): the arrays loadIndexInMemory builds (PQ.java:436-483), for a fast restart */
	public synchronized void saveSnapshot(String filename) throws Exception {
		MmidxNative.saveSnapshot(handle, filename);
	}

	public synchronized void loadSnapshot(String filename) throws Exception {
		MmidxNative.loadSnapshot(handle, filename);
	}
}
