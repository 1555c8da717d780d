package gr.iti.mklab.visual.datastructures;

import gr.iti.mklab.visual.aggregation.AbstractFeatureAggregator;
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

/**
 * Drop-in for {@link IVFPQ}: same constructors and public methods, identity (ids, BDB) stays in
 * Java, vectors-as-codes and search live in the HBM of an MI355X behind libmmidx_hip.so. The five
 * template-method hooks of AbstractSearchStructure (ASS:267, 305, 342, 729, 755) are the only
 * places that differ from IVFPQ.java; each one is a single native call.
 */
public class GpuIVFPQ extends AbstractSearchStructure {

	private final long handle;
	private final int numSubVectors, numProductCentroids, numCoarseCentroids;
	private final Database iidToIvfpqDB;

	public GpuIVFPQ(int vectorLength, int maxNumVectors, boolean readOnly, String BDBEnvHome, int numSubVectors,
			int numProductCentroids, TransformationType transformation, int numCoarseCentroids,
			boolean countSizeOnLoad, int loadCounter, boolean loadIndexInMemory, long cacheSize) throws Exception {
		super(vectorLength, maxNumVectors, readOnly, countSizeOnLoad, loadCounter, loadIndexInMemory, cacheSize);
		if (vectorLength % numSubVectors > 0) { // IVFPQ.java:181-183 (also checked natively)
			throw new Exception("The given number of subvectors is not valid!");
		}
		this.numSubVectors = numSubVectors;
		this.numProductCentroids = numProductCentroids;
		this.numCoarseCentroids = numCoarseCentroids;
		double[] rotation = null;
		if (transformation == TransformationType.RandomRotation) {
			// EJML's createOrthogonal stream cannot be re-derived natively: compute it here, as
			// RandomRotation.java:30-35 does, and hand the D x D matrix over once.
			rotation = org.ejml.ops.RandomMatrices.createOrthogonal(vectorLength, vectorLength,
					new java.util.Random(1)).getData();
		}
		// RandomPermutation(seed = 1, D) is derived natively with the JDK LCG (permutation == null)
		// -Dmmidx.devices=0,1,...,7: the inverted lists are partitioned over these GPUs of the node (mmidx_create_sharded: one
		// RCCL communicator and one native thread per device inside libmmidx_hip.so); otherwise one GPU, -Dmmidx.device=n
		String devs = System.getProperty("mmidx.devices");
		if (devs != null && !devs.trim().isEmpty()) {
			String[] parts = devs.trim().split(",");
			int[] devices = new int[parts.length];
			for (int i = 0; i < parts.length; i++) {
				devices[i] = Integer.parseInt(parts[i].trim());
			}
			handle = MmidxNative.createSharded(MmidxNative.KIND_IVFPQ, vectorLength, numSubVectors, numProductCentroids,
					numCoarseCentroids, transformation.ordinal(), null, rotation, devices);
		} else {
			handle = MmidxNative.create(MmidxNative.KIND_IVFPQ, vectorLength, numSubVectors, numProductCentroids,
					numCoarseCentroids, transformation.ordinal(), null, rotation, Integer.getInteger("mmidx.device", 0));
		}
		createOrOpenBDBEnvAndDbs(BDBEnvHome);
		DatabaseConfig dbConf = new DatabaseConfig();
		dbConf.setReadOnly(readOnly);
		dbConf.setTransactional(transactional);
		dbConf.setAllowCreate(true);
		iidToIvfpqDB = dbEnv.openDatabase(null, "ivfadc", dbConf); // same db name as IVFPQ.java:203
		if (loadIndexInMemory) {
			loadIndexInMemory();
		}
	}

	public GpuIVFPQ(int vectorLength, int maxNumVectors, boolean readOnly, String BDBEnvHome, int numSubVectors,
			int numProductCentroids, TransformationType transformation, int numCoarseCentroids, long cacheSize)
			throws Exception {
		this(vectorLength, maxNumVectors, readOnly, BDBEnvHome, numSubVectors, numProductCentroids, transformation,
				numCoarseCentroids, true, 0, true, cacheSize);
	}

	public void setW(int w) throws Exception { // IVFPQ.java:95-97
		MmidxNative.setW(handle, w);
	}

	public void loadCoarseQuantizer(String filename) throws Exception { // IVFPQ.java:297-300
		double[][] cq = AbstractFeatureAggregator.readQuantizer(filename, numCoarseCentroids, vectorLength);
		double[] flat = new double[numCoarseCentroids * vectorLength];
		for (int i = 0; i < numCoarseCentroids; i++)
			System.arraycopy(cq[i], 0, flat, i * vectorLength, vectorLength);
		MmidxNative.setCoarse(handle, flat);
	}

	public void loadProductQuantizer(String filename) throws Exception { // IVFPQ.java:275-288
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

	/** hook 1 (ASS:267): encode on the GPU, append there, persist the same record as IVFPQ.java:760-792 */
	protected void indexVectorInternal(double[] vector) throws Exception {
		if (vector.length != vectorLength) {
			throw new Exception("The dimensionality of the vector is wrong!");
		}
		int[] cell = new int[1];
		if (numProductCentroids <= 256) { // IVFPQ.java:342-347
			byte[] code = new byte[numSubVectors];
			MmidxNative.addVector(handle, loadCounter, vector, cell, code);
			appendPersistentIndex(cell[0], code);
		} else { // IVFPQ.java:349-354
			short[] code = new short[numSubVectors];
			MmidxNative.addVectorShort(handle, loadCounter, vector, cell, code);
			appendPersistentIndex(cell[0], code);
		}
	}

	public synchronized boolean indexPQCode(String id, int listId, byte[] code) throws Exception { // IVFPQ.java:357-386
		if (numProductCentroids > 256) {
			throw new Exception("Byte is not sufficient to enumerate the centroids of the product quantizer!");
		}
		if (code.length != numSubVectors) {
			throw new Exception("The length of the code is wrong!");
		}
		if (loadCounter >= maxNumVectors) {
			System.out.println("Maximum index capacity reached, no more vectors can be indexed!");
			return false;
		}
		if (isIndexed(id)) {
			System.out.println("Vector '" + id + "' already indexed!");
			return false;
		}
		// native append first: a failure must not leave the id mapped
		MmidxNative.addCodes(handle, 1, new int[] { loadCounter }, new int[] { listId }, code);
		createMapping(id);
		appendPersistentIndex(listId, code);
		loadCounter++;
		return true;
	}

	/** IVFPQ.java:464-497: distance between a query vector and the code of an indexed vector */
	public double computeDistanceIVFADC(double[] qVector, String existingVecId) throws Exception {
		int iid = getInternalId(existingVecId);
		if (iid == -1) {
			throw new Exception("Id does not exist!");
		}
		if (qVector.length != vectorLength) {
			throw new Exception("The dimensionality of the vector is wrong!");
		}
		double[] out = new double[1];
		MmidxNative.distance(handle, qVector, new int[] { iid }, out);
		return out[0];
	}

	/** IVFPQ.java:801-824 (served from the HBM-resident lists; the BDB record holds the same bytes) */
	public byte[] getPQCodeByte(String id) throws Exception {
		int iid = getInternalId(id);
		if (iid == -1) {
			throw new Exception("Id does not exist!");
		}
		if (numProductCentroids > 256) {
			throw new Exception("Call the short variant of the method!");
		}
		byte[] code = new byte[numSubVectors];
		MmidxNative.getCodes(handle, new int[] { iid }, null, code);
		return code;
	}

	/** IVFPQ.java:833-856 */
	public short[] getPQCodeShort(String id) throws Exception {
		int iid = getInternalId(id);
		if (iid == -1) {
			throw new Exception("Id does not exist!");
		}
		if (numProductCentroids <= 256) {
			throw new Exception("Call the short variant of the method!"); // (sic, IVFPQ.java:839)
		}
		short[] code = new short[numSubVectors];
		MmidxNative.getCodesShort(handle, new int[] { iid }, null, code);
		return code;
	}

	/** IVFPQ.java:865-880 */
	public int getInvertedListId(String id) throws Exception {
		int iid = getInternalId(id);
		if (iid == -1) {
			throw new Exception("Id does not exist!");
		}
		int[] cell = new int[1];
		MmidxNative.getCodes(handle, new int[] { iid }, cell, null);
		return cell[0];
	}

	/** hook 2 (ASS:305): one native call; the queue is rebuilt only to satisfy the hook's type */
	protected BoundedPriorityQueue<Result> computeNearestNeighborsInternal(int k, double[] query) throws Exception {
		int[] iids = new int[k];
		double[] dists = new double[k];
		int[] count = new int[1];
		MmidxNative.search(handle, k, 1, query, iids, dists, count);
		BoundedPriorityQueue<Result> nn = new BoundedPriorityQueue<Result>(new Result(), k);
		for (int i = count[0] - 1; i >= 0; i--) // worst first: keeps the queue's tie order
			nn.offer(new Result(iids[i], dists[i]));
		return nn;
	}

	/** hook 3 (ASS:342): IVFPQ.computeKnnIVFSDC returns null in the reference (IVFPQ.java:509-511) */
	protected BoundedPriorityQueue<Result> computeNearestNeighborsInternal(int k, int iid) throws Exception {
		return null;
	}

	/** loadIndexInMemory IVFPQ.java:680-728: stream the BDB records to the native bulk add */
	private void loadIndexInMemory() throws Exception {
		final int B = 1 << 16;
		final boolean bytes = numProductCentroids <= 256;
		int[] iids = new int[B], cells = new int[B];
		byte[] bcodes = bytes ? new byte[B * numSubVectors] : null;
		short[] scodes = bytes ? null : new short[B * numSubVectors];
		int n = 0;
		DatabaseEntry key = new DatabaseEntry(), data = new DatabaseEntry();
		Cursor cursor = iidToIvfpqDB.openCursor(null, null);
		while (cursor.getNext(key, data, LockMode.DEFAULT) == OperationStatus.SUCCESS) {
			TupleInput input = TupleBinding.entryToInput(data);
			cells[n] = input.readInt();
			iids[n] = IntegerBinding.entryToInt(key);
			for (int i = 0; i < numSubVectors; i++) {
				if (bytes)
					bcodes[n * numSubVectors + i] = input.readByte(); // IVFPQ.java:702-705
				else
					scodes[n * numSubVectors + i] = input.readShort(); // IVFPQ.java:707-710
			}
			if (++n == B) {
				if (bytes)
					MmidxNative.addCodes(handle, n, iids, cells, bcodes);
				else
					MmidxNative.addCodesShort(handle, n, iids, cells, scodes);
				n = 0;
			}
		}
		cursor.close();
		if (n > 0) {
			if (bytes)
				MmidxNative.addCodes(handle, n, iids, cells, bcodes); // (the shim reads the first n records only)
			else
				MmidxNative.addCodesShort(handle, n, iids, cells, scodes);
		}
	}

	/**
	 * Native flat snapshot of the in-memory index (SURVEY 8 f1): what {@link #loadIndexInMemory()} builds from the BDB cursor --
	 * list offsets, internal ids and stored codes -- written by the library in one piece (mmidx_save). The BDB environment stays
	 * the system of record for the id map; a restart that finds a snapshot as recent as the database calls
	 * {@link #loadSnapshot(String)} instead of walking the cursor (IVFPQ.java:680-728).
	 */
	public synchronized void saveSnapshot(String filename) throws Exception {
		MmidxNative.saveSnapshot(handle, filename);
	}

	/** fills the (empty) device index from a file written by {@link #saveSnapshot(String)}; quantizers must be loaded first */
	public synchronized void loadSnapshot(String filename) throws Exception {
		MmidxNative.loadSnapshot(handle, filename);
	}

	private void appendPersistentIndex(int listId, byte[] code) { // IVFPQ.java:760-772, unchanged
		TupleOutput output = new TupleOutput();
		output.writeInt(listId);
		for (int i = 0; i < numSubVectors; i++)
			output.writeByte(code[i]);
		DatabaseEntry data = new DatabaseEntry();
		TupleBinding.outputToEntry(output, data);
		DatabaseEntry key = new DatabaseEntry();
		IntegerBinding.intToEntry(loadCounter, key);
		iidToIvfpqDB.put(null, key, data);
	}

	private void appendPersistentIndex(int listId, short[] code) { // IVFPQ.java:780-792, unchanged
		TupleOutput output = new TupleOutput();
		output.writeInt(listId);
		for (int i = 0; i < numSubVectors; i++)
			output.writeShort(code[i]);
		DatabaseEntry data = new DatabaseEntry();
		TupleBinding.outputToEntry(output, data);
		DatabaseEntry key = new DatabaseEntry();
		IntegerBinding.intToEntry(loadCounter, key);
		iidToIvfpqDB.put(null, key, data);
	}

	public void outputItemsPerList() throws Exception { // IVFPQ.java:654-673
		int[] sizes = new int[numCoarseCentroids];
		MmidxNative.listSizes(handle, sizes);
		int max = 0, min = Integer.MAX_VALUE;
		double sum = 0;
		for (int s : sizes) {
			max = Math.max(max, s);
			min = Math.min(min, s);
			sum += s;
		}
		System.out.println("Maximum number of vectors: " + max);
		System.out.println("Minimum number of vectors: " + min);
		System.out.println("Average number of vectors: " + (sum / numCoarseCentroids));
	}

	@Override
	public void outputIndexingTimesInternal() { // hook 4 (ASS:729): empty in the reference; here the device-side timers
		try {
			double[] st = new double[7];
			MmidxNative.stats(handle, st);
			System.out.println("GPU search time since the last report (ms): total " + st[0] + ", coarse " + st[1] + ", scan "
					+ st[2] + ", merge " + st[3] + "; codes scanned " + (long) st[4]);
		} catch (Exception e) {
			System.out.println("GPU statistics unavailable: " + e.getMessage());
		}
	}

	@Override
	public void closeInternal() { // hook 5 (ASS:755)
		iidToIvfpqDB.close();
		MmidxNative.destroy(handle);
	}
}
