package gr.iti.mklab.visual.aggregation;

import gr.iti.mklab.visual.datastructures.MmidxNative;

/**
 * {@link VladAggregatorMultipleVocabularies} on an MI355X: aggregate(double[][]) of
 * VladAggregatorMultipleVocabularies.java:84-101 (per vocabulary VladAggregator.aggregateInternal, VladAggregator.java:56-70,
 * with computeNearestCentroid AbstractFeatureAggregator.java:136-155; power + L2 normalisation, L2 of the concatenation),
 * plus a batch overload for many images. Raw VLAD vectors are bit-exact, normalised ones agree to 1e-12.
 */
public class GpuVladAggregator {

	private final long handle;
	private final int descriptorLength, vectorLength;

	public GpuVladAggregator(double[][][] codebooks, boolean normalizationsOn) throws Exception {
		int[] nc = new int[codebooks.length];
		int total = 0;
		descriptorLength = codebooks[0][0].length;
		for (int v = 0; v < codebooks.length; v++) {
			nc[v] = codebooks[v].length;
			total += nc[v];
		}
		double[] flat = new double[total * descriptorLength];
		int o = 0;
		for (double[][] cb : codebooks)
			for (double[] c : cb) {
				System.arraycopy(c, 0, flat, o, descriptorLength);
				o += descriptorLength;
			}
		handle = MmidxNative.vladCreate(nc, descriptorLength, flat, normalizationsOn, Integer.getInteger("mmidx.device", 0));
		vectorLength = MmidxNative.vladVectorLength(handle);
	}

	public int getVectorLength() {
		return vectorLength;
	}

	public double[] aggregate(double[][] descriptors) throws Exception {
		return aggregate(new double[][][] { descriptors })[0];
	}

	/** one call for a batch of images (the reference runs a thread pool of per-image calls, ImageVectorizer.java:123) */
	public double[][] aggregate(double[][][] images) throws Exception {
		long[] off = new long[images.length + 1];
		for (int i = 0; i < images.length; i++)
			off[i + 1] = off[i] + images[i].length;
		double[] descs = new double[(int) off[images.length] * descriptorLength];
		int o = 0;
		for (double[][] img : images)
			for (double[] d : img) {
				System.arraycopy(d, 0, descs, o, descriptorLength);
				o += descriptorLength;
			}
		double[] out = new double[images.length * vectorLength];
		MmidxNative.vladAggregate(handle, 0L, descriptorLength, vectorLength, off, descs, out);
		double[][] res = new double[images.length][];
		for (int i = 0; i < images.length; i++)
			res[i] = java.util.Arrays.copyOfRange(out, i * vectorLength, (i + 1) * vectorLength);
		return res;
	}

	public void close() {
		MmidxNative.vladDestroy(handle);
	}
}
