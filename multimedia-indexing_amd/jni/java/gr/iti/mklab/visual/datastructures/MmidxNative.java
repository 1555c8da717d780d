package gr.iti.mklab.visual.datastructures;

/**
 * JNI declarations for libmmidx_jni.so (mmidx_jni.c) -> libmmidx_hip.so (include/mmidx.h). The shim checks the length of
 * every array against the handle's dimensions before it touches it; codes travel as byte[] when numProductCentroids <= 256
 * and as short[] otherwise (IVFPQ.java:342-354), and the wrong variant throws.
 */
public final class MmidxNative {
	static {
		System.loadLibrary("mmidx_jni");
	}

	public static final int KIND_PQ = 1, KIND_IVFPQ = 2;

	public static native long create(int kind, int vectorLength, int numSubVectors, int numProductCentroids,
			int numCoarseCentroids, int transformationOrdinal, int[] permutation, double[] rotation, int device)
			throws Exception;

	/**
	 * mmidx_create_sharded: one index over the GPUs listed in devices (inverted list c lives on devices[c % devices.length]); every
	 * other method takes the returned handle unchanged. One JVM drives all of them, as the reference's caller holds the whole
	 * index in one process.
	 */
	public static native long createSharded(int kind, int vectorLength, int numSubVectors, int numProductCentroids,
			int numCoarseCentroids, int transformationOrdinal, int[] permutation, double[] rotation, int[] devices)
			throws Exception;

	public static native void destroy(long handle);

	public static native void setCoarse(long handle, double[] flatCoarse) throws Exception;

	public static native void setPq(long handle, double[] flatProductQuantizer) throws Exception;

	public static native void setW(long handle, int w) throws Exception;

	/** mmidx_add_vectors for one vector; cellOut[0] and codeOut[numSubVectors] receive the record */
	public static native void addVector(long handle, int iid, double[] vector, int[] cellOut, byte[] codeOut)
			throws Exception;

	public static native void addVectorShort(long handle, int iid, double[] vector, int[] cellOut, short[] codeOut)
			throws Exception;

	/** mmidx_add_codes: n records; cells == null for PQ */
	public static native void addCodes(long handle, int n, int[] iids, int[] cells, byte[] codes) throws Exception;

	public static native void addCodesShort(long handle, int n, int[] iids, int[] cells, short[] codes) throws Exception;

	/** mmidx_get_codes: list ids and stored codes of iids.length internal ids; either output may be null */
	public static native void getCodes(long handle, int[] iids, int[] cellsOut, byte[] codesOut) throws Exception;

	public static native void getCodesShort(long handle, int[] iids, int[] cellsOut, short[] codesOut) throws Exception;

	/** mmidx_distance: computeDistanceIVFADC for iids.length (query, internal id) pairs, queries row-major */
	public static native void distance(long handle, double[] queries, int[] iids, double[] out) throws Exception;

	public static native void search(long handle, int k, int nq, double[] queries, int[] iidOut, double[] distOut,
			int[] countOut) throws Exception;

	/** mmidx_search_sdc: queries are internal ids (PQ.computeKnnSDC, PQ.java:334-374) */
	public static native void searchSdc(long handle, int k, int nq, int[] queryIids, int[] iidOut, double[] distOut,
			int[] countOut) throws Exception;

	public static native void listSizes(long handle, int[] out) throws Exception;

	/** mmidx_save / mmidx_load (ABI 8): the native flat snapshot -- list offsets, iids and stored codes as loadIndexInMemory builds them */
	public static native void saveSnapshot(long handle, String path) throws Exception;

	public static native void loadSnapshot(long handle, String path) throws Exception;

	/** mmidx_get_stats as {total_ms, coarse_ms, scan_ms, merge_ms, scan_codes, scan_launches, tie_fallbacks} */
	public static native void stats(long handle, double[] out7) throws Exception;

	public static native void setProfiling(long handle, int mode) throws Exception;

	/* ---- Linear (mmidx_linear_*) ---- */
	public static native long linearCreate(int vectorLength, long maxNumVectors, int device) throws Exception;

	public static native void linearDestroy(long handle);

	public static native void linearAdd(long handle, int n, int vectorLength, double[] flatVectors) throws Exception;

	public static native void linearSearch(long handle, int k, int nq, int vectorLength, double[] queries, int[] iidOut,
			double[] distOut, int[] countOut) throws Exception;

	/* ---- front end (mmidx_pca_*, mmidx_vlad_*, mmidx_vectorize) ---- */
	public static native long pcaCreate(int numComponents, int sampleSize, boolean whitening, double[] means,
			double[] eigenvalues, double[] components, int device) throws Exception;

	public static native void pcaDestroy(long pca);

	public static native void pcaProject(long pca, int n, int sampleSize, int numComponents, double[] samples,
			double[] projected) throws Exception;

	public static native long vladCreate(int[] numCentroids, int descriptorLength, double[] codebooks,
			boolean normalizationsOn, int device) throws Exception;

	public static native void vladDestroy(long vlad);

	public static native int vladVectorLength(long vlad) throws Exception;

	/** pca == 0: aggregate only (outLen = VLAD length); else aggregate + sampleToEigenSpace (outLen = components) */
	public static native void vladAggregate(long vlad, long pca, int descriptorLength, int outLen, long[] descOff,
			double[] descriptors, double[] out) throws Exception;

	private MmidxNative() {
	}
}
