package gr.iti.mklab.visual.dimreduction;

import gr.iti.mklab.visual.datastructures.MmidxNative;

import java.io.BufferedReader;
import java.io.FileReader;

/**
 * Apply side of {@link PCA} on an MI355X: loadPCAFromFile (PCA.java:257-318) + sampleToEigenSpace (PCA.java:188-208), plus a
 * batch overload (one f64-MFMA GEMM for n samples). Learning the basis (addSample / computeBasis, EJML SVD) stays with PCA.
 * Results agree with EJML to 1e-12 relative to the row norm, not bit for bit (summation order).
 */
public class GpuPCA {

	private final int numComponents, sampleSize;
	private final boolean doWhitening;
	private long handle;

	public GpuPCA(int numComponents, int numTrainingSamples, int sampleSize, boolean doWhitening) { // PCA.java:75-93
		this.numComponents = numComponents;
		this.sampleSize = sampleSize;
		this.doWhitening = doWhitening;
	}

	public void loadPCAFromFile(String PCAFileName) throws Exception { // PCA.java:257-318
		BufferedReader in = new BufferedReader(new FileReader(PCAFileName));
		String[] meanString = in.readLine().trim().split(" ");
		if (meanString.length != sampleSize) {
			in.close();
			throw new Exception("Means line is wrong!");
		}
		double[] means = new double[sampleSize];
		for (int i = 0; i < sampleSize; i++)
			means[i] = Double.parseDouble(meanString[i]);
		String line = in.readLine();
		double[] eig = null;
		if (doWhitening) {
			String[] eigString = line.trim().split(" ");
			eig = new double[numComponents];
			for (int i = 0; i < numComponents; i++)
				eig[i] = Double.parseDouble(eigString[i]);
		}
		double[] vt = new double[numComponents * sampleSize];
		for (int i = 0; i < numComponents; i++) {
			String[] comp = in.readLine().trim().split(" ");
			for (int j = 0; j < sampleSize; j++)
				vt[i * sampleSize + j] = Double.parseDouble(comp[j]);
		}
		in.close();
		// the whitening matrix diag(eig^-0.5) is folded into V_t natively, as PCA.java:283-313 does
		handle = MmidxNative.pcaCreate(numComponents, sampleSize, doWhitening, means, eig, vt,
				Integer.getInteger("mmidx.device", 0));
	}

	public double[] sampleToEigenSpace(double[] sampleData) throws Exception { // PCA.java:188-208
		if (handle == 0) {
			throw new Exception("PCA is not correctly initialized!");
		}
		if (sampleData.length != sampleSize) {
			throw new IllegalArgumentException("Unexpected vector length!");
		}
		double[] out = new double[numComponents];
		MmidxNative.pcaProject(handle, 1, sampleSize, numComponents, sampleData, out);
		return out;
	}

	/** n samples row-major -> n projected vectors row-major */
	public double[] samplesToEigenSpace(double[] samples, int n) throws Exception {
		double[] out = new double[n * numComponents];
		MmidxNative.pcaProject(handle, n, sampleSize, numComponents, samples, out);
		return out;
	}

	long nativeHandle() {
		return handle;
	}

	public void close() {
		if (handle != 0)
			MmidxNative.pcaDestroy(handle);
		handle = 0;
	}
}
