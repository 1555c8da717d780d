package gr.iti.mklab.visual.datastructures;

/** JNI declarations for libmmidx_jni.so (mmidx_jni.c) -> libmmidx_hip.so (include/mmidx.h). */
final class MmidxNative {
	static {
		System.loadLibrary("mmidx_jni");
	}

	static final int KIND_PQ = 1, KIND_IVFPQ = 2;

	static native long create(int kind, int vectorLength, int numSubVectors, int numProductCentroids,
			int numCoarseCentroids, int transformationOrdinal, int[] permutation, double[] rotation, int device)
			throws Exception;

	static native void destroy(long handle);

	static native void setCoarse(long handle, double[] flatCoarse) throws Exception;

	static native void setPq(long handle, double[] flatProductQuantizer) throws Exception;

	static native void setW(long handle, int w) throws Exception;

	static native void addVector(long handle, int iid, double[] vector, int[] cellOut, byte[] codeOut)
			throws Exception;

	static native void addCodes(long handle, int n, int[] iids, int[] cells, byte[] codes) throws Exception;

	static native void search(long handle, int k, int nq, double[] queries, int[] iidOut, double[] distOut,
			int[] countOut) throws Exception;

	/** mmidx_search_sdc: queries are internal ids (PQ.computeKnnSDC, PQ.java:334-374) */
	static native void searchSdc(long handle, int k, int nq, int[] queryIids, int[] iidOut, double[] distOut,
			int[] countOut) throws Exception;

	static native void listSizes(long handle, int[] out) throws Exception;

	/* ---- Linear (mmidx_linear_*) ---- */
	static native long linearCreate(int vectorLength, long maxNumVectors, int device) throws Exception;

	static native void linearDestroy(long handle);

	static native void linearAdd(long handle, int n, double[] flatVectors) throws Exception;

	static native void linearSearch(long handle, int k, int nq, double[] queries, int[] iidOut, double[] distOut,
			int[] countOut) throws Exception;

	private MmidxNative() {
	}
}
