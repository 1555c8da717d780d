// hf_dev.hip -- development harness (scratch, not product): K3h (k_scan_hist) against K3hf (k_scan_hf) on a synthetic cfg4-shaped
// pass A: 8192 lists x 12207 random codes, 16384 queries -> random nearest lists.  Compares the pools / thresholds the two kernels
// publish (must be identical as sets) and times both.
// build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -I../../include -o hf_dev hf_dev.hip
#include "../../multimedia-indexing_amd/csrc/mmidx_kernels.h"
#include "../../multimedia-indexing_amd/csrc/mmidx_scan_q.h"

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <random>
#include <vector>

#define CHECK(x)                                                                          \
    do {                                                                                  \
        hipError_t e_ = (x);                                                              \
        if (e_ != hipSuccess) {                                                           \
            printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); \
            exit(1);                                                                      \
        }                                                                                 \
    } while (0)

__global__ void k_fill_codes(u32 *p, size_t nwords, u32 seed) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < nwords; i += (size_t)gridDim.x * blockDim.x) {
        u32 x = (u32)i * 2654435761u + seed;
        x ^= x >> 16;
        x *= 0x7feb352du;
        x ^= x >> 15;
        x *= 0x846ca68bu;
        x ^= x >> 16;
        p[i] = x;
    }
}

int main(int argc, char **argv) {
#ifndef HD
#define HD 128
#endif
#ifndef HK1
#define HK1 101
#endif
#ifndef HRES
#define HRES 0.15
#endif
    const int D = HD, M = 16, ks = 256, dsub = HD / 16, C = 8192, K1 = HK1;
    const int nq = argc > 1 ? atoi(argv[1]) : 16384;
    const int avg = argc > 2 ? atoi(argv[2]) : 12207;
    const int poolq = 1024;
    std::mt19937_64 rng(7);
    std::normal_distribution<double> N01(0.0, 1.0);
    // lists of slightly different lengths
    std::vector<int64_t> off(C + 1, 0);
    for (int c = 0; c < C; c++) off[c + 1] = off[c] + avg + (int)(rng() % 201) - 100;
    const int64_t n = off[C];
    std::vector<double> coarse((size_t)C * D), Q((size_t)nq * D), pqT((size_t)M * dsub * ks);
    for (auto &v : coarse) v = N01(rng);
    for (auto &v : pqT) v = 0.15 * N01(rng);
    std::vector<int32_t> cells(nq);
    for (int q = 0; q < nq; q++) {
        cells[q] = (int32_t)(rng() % C);
        for (int j = 0; j < D; j++) Q[(size_t)q * D + j] = coarse[(size_t)cells[q] * D + j] + HRES * N01(rng);
    }
    double *d_coarse, *d_Q, *d_pqT;
    int32_t *d_cells, *d_fb;
    int64_t *d_off;
    unsigned char *d_codes;
    u64 *d_T, *d_pkey, *d_pval;
    u32 *d_pcnt;
    CHECK(hipMalloc((void **)&d_coarse, coarse.size() * 8));
    CHECK(hipMalloc((void **)&d_Q, Q.size() * 8));
    CHECK(hipMalloc((void **)&d_pqT, pqT.size() * 8));
    CHECK(hipMalloc((void **)&d_cells, nq * 4));
    CHECK(hipMalloc((void **)&d_off, (C + 1) * 8));
    CHECK(hipMalloc((void **)&d_codes, (size_t)n * M + 64));
    CHECK(hipMalloc((void **)&d_T, (size_t)nq * 8));
    CHECK(hipMalloc((void **)&d_pcnt, (size_t)nq * 4));
    CHECK(hipMalloc((void **)&d_pkey, (size_t)nq * poolq * 8));
    CHECK(hipMalloc((void **)&d_pval, (size_t)nq * poolq * 8));
    CHECK(hipMalloc((void **)&d_fb, (size_t)(2 * nq + 4) * 4));
    CHECK(hipMemcpy(d_coarse, coarse.data(), coarse.size() * 8, hipMemcpyHostToDevice));
    CHECK(hipMemcpy(d_Q, Q.data(), Q.size() * 8, hipMemcpyHostToDevice));
    CHECK(hipMemcpy(d_pqT, pqT.data(), pqT.size() * 8, hipMemcpyHostToDevice));
    CHECK(hipMemcpy(d_cells, cells.data(), nq * 4, hipMemcpyHostToDevice));
    CHECK(hipMemcpy(d_off, off.data(), (C + 1) * 8, hipMemcpyHostToDevice));
    hipLaunchKernelGGL(k_fill_codes, dim3(4096), dim3(256), 0, nullptr, (u32 *)d_codes, (size_t)n * M / 4, 99u);
    CHECK(hipDeviceSynchronize());

    ScanParams P{};
    P.Q = d_Q;
    P.coarse = d_coarse;
    P.pqT = d_pqT;
    P.cells = d_cells;
    P.list_off = d_off;
    P.codes = d_codes;
    P.T = d_T;
    P.pool_cnt = d_pcnt;
    P.pool_key = d_pkey;
    P.pool_val = d_pval;
    P.D = D;
    P.m = M;
    P.ks = ks;
    P.dsub = dsub;
    P.w = 1;
    P.transform = 0;
    P.ivf = 1;
    P.n_items = nq;
    P.rank_lo = 0;
    P.nrank = 1;
    P.chunk = 1 << 24;
    P.fb_count = (u32 *)d_fb;
    P.fb_items = d_fb + 4;
    P.fb_ch = d_fb + 4 + nq;
    P.code_lo = 0;
    P.code_hi = 0x7fffffff;
    P.K1 = K1;
    P.poolq = poolq;

    struct Res {
        std::vector<u64> T, key, val;
        std::vector<u32> cnt;
        int fb, c1, c2, c3;
        float ms;
    };
    auto reset = [&]() {
        CHECK(hipMemset(d_T, 0xFF, (size_t)nq * 8));
        CHECK(hipMemset(d_pcnt, 0, (size_t)nq * 4));
        CHECK(hipMemset(d_fb, 0, 16));
    };
    auto collect = [&](Res &r) {
        r.T.resize(nq);
        r.cnt.resize(nq);
        r.key.resize((size_t)nq * poolq);
        r.val.resize((size_t)nq * poolq);
        CHECK(hipMemcpy(r.T.data(), d_T, (size_t)nq * 8, hipMemcpyDeviceToHost));
        CHECK(hipMemcpy(r.cnt.data(), d_pcnt, (size_t)nq * 4, hipMemcpyDeviceToHost));
        CHECK(hipMemcpy(r.key.data(), d_pkey, (size_t)nq * poolq * 8, hipMemcpyDeviceToHost));
        CHECK(hipMemcpy(r.val.data(), d_pval, (size_t)nq * poolq * 8, hipMemcpyDeviceToHost));
        int32_t c4[4];
        CHECK(hipMemcpy(c4, d_fb, 16, hipMemcpyDeviceToHost));
        r.fb = c4[0];
        r.c1 = c4[1];
        r.c2 = c4[2];
        r.c3 = c4[3];
    };
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0));
    CHECK(hipEventCreate(&e1));
    auto timeit = [&](auto launch, Res &r) {
        reset();
        launch();
        CHECK(hipDeviceSynchronize());
        collect(r);
        float best = 1e9f;
        for (int i = 0; i < 5; i++) {
            reset();
            CHECK(hipEventRecord(e0, nullptr));
            launch();
            CHECK(hipEventRecord(e1, nullptr));
            CHECK(hipEventSynchronize(e1));
            float ms;
            CHECK(hipEventElapsedTime(&ms, e0, e1));
            best = std::min(best, ms);
        }
        r.ms = best;
    };
    // K3h as the library launches it: four blocks per CU
    Res rh, rf;
    {
        const size_t fixed = (size_t)M * ks * 8 + (size_t)D * 8 + 2 * MMIDX_HWV * 8 + 16 + MMIDX_HB * 4 + MMIDX_HCNT * 4;
        int64_t room = (int64_t)(160 * 1024 / 4) - 256 - (int64_t)fixed;
        int cap = (int)std::min<int64_t>(4096, std::max<int64_t>(768, room / 4)) & ~15;
        const size_t lds = fixed + (size_t)cap * 4;
        ScanParams PH = P;
        PH.cap = cap;
        CHECK(hipFuncSetAttribute((const void *)k_scan_hist<16, 256, 256>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        timeit([&]() { hipLaunchKernelGGL((k_scan_hist<16, 256, 256>), dim3(nq), dim3(256), lds, nullptr, PH); }, rh);
        printf("K3h : %.3f ms  (cap %d, lds %zu, %d handed back)  %.0f GB/s of codes\n", rh.ms, cap, lds, rh.fb, (double)nq * avg * 16 / rh.ms / 1e6);
    }
    {
        // K3q: the queries sorted by nearest cell, groups of <= 4 per list
        std::vector<int32_t> order(nq);
        for (int q = 0; q < nq; q++) order[q] = q;
        std::stable_sort(order.begin(), order.end(), [&](int a, int b) { return cells[a] < cells[b]; });
        std::vector<int4> gdesc;
        for (int i = 0; i < nq;) {
            int j = i;
            while (j < nq && cells[order[j]] == cells[order[i]]) j++;
            for (int o = i; o < j; o += MMIDX_Q_G) gdesc.push_back(make_int4(cells[order[i]], o, std::min(MMIDX_Q_G, j - o), 0));
            i = j;
        }
        const int ngr = (int)gdesc.size();
        std::vector<double> pq((size_t)M * ks * dsub), pqstat((size_t)M * dsub + M, 0.0);
        for (int s = 0; s < M; s++)
            for (int j = 0; j < ks; j++) {
                double nn = 0.0;
                for (int t = 0; t < dsub; t++) {
                    const double v = pqT[((size_t)s * dsub + t) * ks + j];
                    pq[((size_t)s * ks + j) * dsub + t] = v;
                    pqstat[(size_t)s * dsub + t] += v / ks;
                    nn += v * v;
                }
                pqstat[(size_t)M * dsub + s] += nn / ks;
            }
        int32_t *d_order, *d_ng;
        int4 *d_gdesc;
        double *d_pq, *d_pqstat;
        CHECK(hipMalloc((void **)&d_order, nq * 4));
        CHECK(hipMalloc((void **)&d_ng, 4));
        CHECK(hipMalloc((void **)&d_gdesc, (size_t)ngr * 16));
        CHECK(hipMalloc((void **)&d_pq, pq.size() * 8));
        CHECK(hipMalloc((void **)&d_pqstat, pqstat.size() * 8));
        CHECK(hipMemcpy(d_order, order.data(), nq * 4, hipMemcpyHostToDevice));
        CHECK(hipMemcpy(d_ng, &ngr, 4, hipMemcpyHostToDevice));
        CHECK(hipMemcpy(d_gdesc, gdesc.data(), (size_t)ngr * 16, hipMemcpyHostToDevice));
        CHECK(hipMemcpy(d_pq, pq.data(), pq.size() * 8, hipMemcpyHostToDevice));
        CHECK(hipMemcpy(d_pqstat, pqstat.data(), pqstat.size() * 8, hipMemcpyHostToDevice));
        QParams QP{};
        QP.S = P;
        QP.S.order = d_order;
        QP.gdesc = d_gdesc;
        QP.n_groups = d_ng;
        QP.pq = d_pq;
        QP.pqstat = d_pqstat;
        {
            std::vector<float> t32(pq.size());
            double r2 = 0.0;
            for (int s = 0; s < M; s++) {
                double mx = 0.0;
                for (int j = 0; j < ks; j++) {
                    double nn = 0.0;
                    for (int t = 0; t < dsub; t++) {
                        const double v = pq[((size_t)s * ks + j) * dsub + t];
                        t32[(((size_t)s * (dsub / 2) + t / 2) * 256 + j) * 2 + (t & 1)] = (float)v;
                        nn += v * v;
                    }
                    mx = std::max(mx, nn);
                }
                r2 += mx;
            }
            float *d_t32;
            CHECK(hipMalloc((void **)&d_t32, t32.size() * 4));
            CHECK(hipMemcpy(d_t32, t32.data(), t32.size() * 4, hipMemcpyHostToDevice));
            QP.pqT32 = d_t32;
            QP.pmax = std::sqrt(r2) * (1.0 + 1e-12);
        }
        unsigned long long *d_tim;
        CHECK(hipMalloc((void **)&d_tim, 64));
        CHECK(hipMemset(d_tim, 0, 64));
        QP.timing = d_tim;
        const QLds L(M, D);
        CHECK(hipFuncSetAttribute((const void *)k_scan_q<16, HD / 16>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)L.total));
        hipFuncAttributes fa{};
        CHECK(hipFuncGetAttributes(&fa, (const void *)k_scan_q<16, HD / 16>));
        timeit([&]() { hipLaunchKernelGGL((k_scan_q<16, HD / 16>), dim3((ngr + 14) & ~7), dim3(256), L.total, nullptr, QP); }, rf);
        int occ = 0;
        CHECK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, (const void *)k_scan_q<16, HD / 16>, 256, L.total));
        printf("occupancy: %d blocks per CU\n", occ);
        printf("K3q : %.3f ms  (%d groups, lds %zu, static lds %zu, %d VGPRs, %d queries handed back)  %.0f GB/s of codes (algorithmic: every query's list)\n", rf.ms, ngr, L.total,
               (size_t)fa.sharedSizeBytes, fa.numRegs, rf.fb, (double)nq * avg * 16 / rf.ms / 1e6);
        unsigned long long tim[8];
        CHECK(hipMemcpy(tim, d_tim, 64, hipMemcpyDeviceToHost));
        printf("      cycles per block (thread 0, 6 launches): residual+scale %.0f, table %.0f, scan %.0f, selection %.0f, exact sums %.0f, publish %.0f\n", tim[0] / 6.0 / ngr, tim[1] / 6.0 / ngr,
               tim[2] / 6.0 / ngr, tim[3] / 6.0 / ngr, tim[4] / 6.0 / ngr, tim[5] / 6.0 / ngr);
    }
    // check K3hf against brute force on the host (the first NV queries): T valid and the pool = exactly the codes with d <= T
    const int NV = std::min(nq, argc > 4 ? atoi(argv[4]) : 256);
    std::vector<unsigned char> hcodes;
    long long bad = 0, tot = 0, under = 0, handed = 0;
    for (int q = 0; q < NV; q++) {
        const int cell = cells[q];
        const int64_t len = off[cell + 1] - off[cell];
        hcodes.resize((size_t)len * M);
        CHECK(hipMemcpy(hcodes.data(), d_codes + (size_t)off[cell] * M, (size_t)len * M, hipMemcpyDeviceToHost));
        std::vector<double> lut((size_t)M * ks);
        for (int s = 0; s < M; s++)
            for (int j = 0; j < ks; j++) {
                double acc = 0.0;
                for (int t = 0; t < dsub; t++) {
                    const double r = coarse[(size_t)cell * D + s * dsub + t] - Q[(size_t)q * D + s * dsub + t];
                    const double df = r - pqT[((size_t)s * dsub + t) * ks + j];
                    acc += df * df;
                }
                lut[(size_t)s * ks + j] = acc;
            }
        std::vector<std::pair<u64, u64>> all((size_t)len);
        for (int64_t i = 0; i < len; i++) {
            double d = lut[hcodes[(size_t)i * M]];
            for (int s = 1; s < M; s++) d += lut[(size_t)s * ks + hcodes[(size_t)i * M + s]];
            u64 k;
            memcpy(&k, &d, 8);
            all[(size_t)i] = {k, (u64)i};
        }
        std::sort(all.begin(), all.end());
        const u64 T = rf.T[q];
        if (T == MMIDX_KEY_MAX && rf.cnt[q] == 0) { handed++; continue; }
        size_t nle = 0;
        while (nle < all.size() && all[nle].first <= T) nle++;
        const u32 c = std::min<u32>(rf.cnt[q], poolq);
        std::vector<std::pair<u64, u64>> b(c);
        for (u32 i = 0; i < c; i++) b[i] = {rf.key[(size_t)q * poolq + i], rf.val[(size_t)q * poolq + i] & 0xFFFFFFFFull};
        std::sort(b.begin(), b.end());
        bool ok = nle >= (size_t)K1 && b.size() == nle;
        for (size_t i = 0; ok && i < nle; i++) ok = b[i] == all[i];
        if (!ok) {
            if (bad < 5) printf("query %d: T rank %zu (K1 %d), pool %zu entries, K3h T %llx, K3hf T %llx, exact K1-th %llx\n", q, nle, K1, b.size(), rh.T[q], T, all[K1 - 1].first);
            bad++;
        }
        tot += c;
        under += (long long)nle;
    }
    printf("%lld of the checked queries were handed back (not checked here); ", handed);
    printf("%s: %lld of %d checked queries wrong; %.1f pool entries per query, %.1f codes at or under T (K1 = %d; K3h: %.1f pool entries)\n", bad ? "MISMATCH" : "correct", bad, NV,
           (double)tot / NV, (double)under / NV, K1, [&]() { double t = 0; for (int q = 0; q < NV; q++) t += rh.cnt[q]; return t / NV; }());
    return bad ? 1 : 0;
}
