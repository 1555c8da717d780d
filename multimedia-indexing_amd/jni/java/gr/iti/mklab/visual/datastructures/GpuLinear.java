package gr.iti.mklab.visual.datastructures;

import gr.iti.mklab.visual.utilities.Result;

import com.aliasi.util.BoundedPriorityQueue;
import com.sleepycat.bind.tuple.IntegerBinding;
import com.sleepycat.bind.tuple.TupleBinding;
import com.sleepycat.bind.tuple.TupleInput;
import com.sleepycat.bind.tuple.TupleOutput;
import com.sleepycat.je.Cursor;
import com.sleepycat.je.Database;
import com.sleepycat.je.DatabaseConfig;
import com.sleepycat.je.DatabaseEntry;
import com.sleepycat.je.OperationStatus;

/**
 * Drop-in for {@link Linear} (exhaustive exact search, Linear.java:138-163) over libmmidx_hip.so: same constructors, the
 * BDB store ("vlad" database, one record of vectorLength doubles per iid, Linear.java:225-236) stays in Java, the
 * in-memory vectors and the search live on the GPU (mmidx_linear_*).  Not compiled here (no JDK in the build image).
 */
public class GpuLinear extends AbstractSearchStructure {

	private final long handle;
	private final Database iidToVectorDB;

	public GpuLinear(int vectorLength, int maxNumVectors, boolean readOnly, String BDBEnvHome, boolean loadIndexInMemory,
			boolean countSizeOnLoad, int loadCounter) throws Exception {
		super(vectorLength, maxNumVectors, readOnly, countSizeOnLoad, loadCounter, loadIndexInMemory);
		handle = MmidxNative.linearCreate(vectorLength, maxNumVectors, Integer.getInteger("mmidx.device", 0));
		createOrOpenBDBEnvAndDbs(BDBEnvHome);
		DatabaseConfig dbConf = new DatabaseConfig();
		dbConf.setReadOnly(readOnly);
		dbConf.setTransactional(transactional);
		dbConf.setAllowCreate(true);
		iidToVectorDB = dbEnv.openDatabase(null, "vlad", dbConf); // Linear.java:78
		if (loadIndexInMemory) {
			loadIndexInMemory();
		}
	}

	public GpuLinear(int vectorLength, int maxNumVectors, boolean readOnly, String BDBEnvHome) throws Exception {
		this(vectorLength, maxNumVectors, readOnly, BDBEnvHome, true, true, 0);
	}

	protected void indexVectorInternal(double[] vector) throws Exception { // Linear.java:111-122
		if (vector.length != vectorLength) {
			throw new Exception("The dimensionality of the vector is wrong!");
		}
		TupleOutput output = new TupleOutput();
		for (int i = 0; i < vectorLength; i++) {
			output.writeDouble(vector[i]);
		}
		DatabaseEntry data = new DatabaseEntry();
		TupleBinding.outputToEntry(output, data);
		DatabaseEntry key = new DatabaseEntry();
		IntegerBinding.intToEntry(loadCounter, key);
		iidToVectorDB.put(null, key, data);
		if (loadIndexInMemory) {
			MmidxNative.linearAdd(handle, 1, vectorLength, vector);
		}
	}

	protected BoundedPriorityQueue<Result> computeNearestNeighborsInternal(int k, double[] queryVector) throws Exception {
		int[] iids = new int[k];
		double[] dists = new double[k];
		int[] count = new int[1];
		MmidxNative.linearSearch(handle, k, 1, vectorLength, queryVector, iids, dists, count); // -> mmidx_linear_search
		BoundedPriorityQueue<Result> nn = new BoundedPriorityQueue<Result>(new Result(), k);
		for (int i = count[0] - 1; i >= 0; i--)
			nn.offer(new Result(iids[i], dists[i])); // worst first keeps the tie order
		return nn;
	}

	protected BoundedPriorityQueue<Result> computeNearestNeighborsInternal(int k, int iid) throws Exception {
		return computeNearestNeighborsInternal(k, getVector(iid)); // Linear.java:181-186
	}

	public double[] getVector(int iid) { // Linear.java:253-280 (disk-based branch)
		DatabaseEntry key = new DatabaseEntry();
		IntegerBinding.intToEntry(iid, key);
		DatabaseEntry data = new DatabaseEntry();
		if (iidToVectorDB.get(null, key, data, null) != OperationStatus.SUCCESS) {
			return null;
		}
		TupleInput input = TupleBinding.entryToInput(data);
		double[] vector = new double[vectorLength];
		for (int i = 0; i < vectorLength; i++) {
			vector[i] = input.readDouble();
		}
		return vector;
	}

	private void loadIndexInMemory() throws Exception { // Linear.java:191-222, batched
		final int B = 4096;
		double[] buf = new double[B * vectorLength];
		int n = 0, counter = 0;
		DatabaseEntry key = new DatabaseEntry(), data = new DatabaseEntry();
		Cursor cursor = iidToVectorDB.openCursor(null, null);
		while (cursor.getNext(key, data, null) == OperationStatus.SUCCESS && counter < maxNumVectors) {
			TupleInput input = TupleBinding.entryToInput(data);
			for (int i = 0; i < vectorLength; i++) {
				buf[n * vectorLength + i] = input.readDouble();
			}
			n++;
			counter++;
			if (n == B) {
				MmidxNative.linearAdd(handle, n, vectorLength, buf);
				n = 0;
			}
		}
		if (n > 0) {
			MmidxNative.linearAdd(handle, n, vectorLength, java.util.Arrays.copyOf(buf, n * vectorLength));
		}
		cursor.close();
	}

	protected void outputIndexingTimesInternal() {
	}

	protected void closeInternal() {
		iidToVectorDB.close();
		MmidxNative.linearDestroy(handle);
	}
}
