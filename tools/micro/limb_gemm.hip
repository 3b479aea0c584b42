// limb_gemm.hip -- go / no-go measurement for an exact fixed-point GEMM on the i8 MFMA (v_mfma_i32_32x32x32_i8):
// both operands are rows of 4 signed 8-bit limbs per element (value = l0*2^24 + l1*2^16 + l2*2^8 + l3), the kernel forms
// the ten limb products of weight >= 2^24 in four i32 accumulator sets and combines them exactly.
//   hipcc --offload-arch=gfx950 -O3 -o _limb limb_gemm.hip && ./_limb
// Layout of a limb matrix with R rows: plane (c, l) for the 16-byte k chunk c and limb l, each plane R x 16 bytes:
//   byte address = ((c * 4 + l) * R + row) * 16 + (k % 16)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>
#include <vector>
#include <cmath>
#ifndef MODE
#define MODE 0
#endif

typedef int v4i __attribute__((ext_vector_type(4)));
typedef int v16i __attribute__((ext_vector_type(16)));

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

constexpr int TM = 128, TN = 128;
constexpr int NBUF = 4;                 // LDS ring: one 32-byte k step (2 chunks x 4 limbs x 2 operands = 32 KB) per slot
constexpr int SLOT = 32768, OPB = 16384;


// C[m][n] = sum_k A[m][k] * B[n][k] over the kept limb products, as fp32.  Persistent workgroups: workgroup w takes the
// tiles w, w + gridDim, ... of an XCD-aware order and runs their k steps as ONE stream through the LDS ring (the first
// steps of the next tile are in flight while this one finishes and stores).
__global__ void __launch_bounds__(256)
k_limb_gemm(const int8_t *__restrict__ A, long RA, const int8_t *__restrict__ Bm, long RB, int Dp, float *__restrict__ out, int) {
    constexpr int mode = MODE;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave & 1, wn = wave >> 1;
    const int NT = (int)(RB / TN);
    const long MT = RA / TM;
    const long ntiles = MT * NT;
    // tile order: XCD x (workgroup id mod 8) owns the row tiles x, x + 8, ...; inside an XCD consecutive tiles walk the
    // column tiles of one row tile (they share the A slab while it is hot in that L2)
    const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3, per_xcd = gridDim.x >> 3;
    auto tile_of = [&](long q, long &m0, int &n0) -> bool {      // q-th tile of this workgroup
        const long idx = slot + q * per_xcd;
        const long mt = (idx / NT) * 8 + xcd;
        m0 = mt * TM;
        n0 = (int)(idx % NT) * TN;
        return mt < MT;
    };
    v16i acc[2][2][4];
    const int nst = (mode == 2) ? 4 : Dp / 32;
    const int wu = __builtin_amdgcn_readfirstlane(wave);
    const int op = wu >> 1;
    const long R = op ? RB : RA;
    const long pl = R * 16, sstride = 8 * R * 16;
    const unsigned dbase = (unsigned)(size_t)smem + op * OPB + (4 * (wu & 1)) * 2048;
    const unsigned voff = lane * 16;
    auto base_of = [&](long m0, int n0) { return (op ? Bm + (long)n0 * 16 : A + m0 * 16) + (long)(4 * (wu & 1)) * R * 16; };
    // one plane (two 1 KB pieces: rows 0..63 and 64..127 of the tile) of a stage: p = its first byte, st its ring position
    auto issue1 = [&](const int8_t *p, int st, int g) {
        const int8_t *pg = p + g * pl;
        const unsigned d = dbase + (st % NBUF) * SLOT + g * 2048;
        asm volatile(
            "s_mov_b32 m0, %[d]\n\ts_nop 0\n\t"
            "global_load_lds_dwordx4 %[vo], %[p]\n\t"
            "global_load_lds_dwordx4 %[vo], %[p] offset:1024\n\t"
            :
            : [vo] "v"(voff), [p] "s"(pg), [d] "s"(d)
            : "memory", "m0");
    };
    const int r32 = lane & 31, kh = lane >> 5;
    long m0, m0n;
    int n0, n0n;
    if (!tile_of(0, m0, n0)) return;
    const int8_t *pcur = base_of(m0, n0), *pnext = pcur;
    // a k step: the MFMAs of stage st (fragments in a, b), the fragment reads of stage st + 1 (into an, bn) and the loads
    // of stage st + 4 (of the next tile past the end of this one) into the slot stage st has just left, in four groups
    auto step = [&](int st, bool has_next, const v4i (&a)[2][4], const v4i (&b)[2][4], v4i (&an)[2][4], v4i (&bn)[2][4]) {
        const char *base = smem + ((st + 1) % NBUF) * SLOT;
        const bool load = (st + 4 < nst) || has_next;
        const int8_t *p = (st + 4 < nst) ? pcur + (st + 4) * sstride : pnext + (st + 4 - nst) * sstride;
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            if (load && !(mode & 16)) issue1(p, st, g);
            if (!(mode & 32))
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                an[t][g] = *reinterpret_cast<const v4i *>(base + (kh * 4 + g) * 2048 + (64 * wm + 32 * t + r32) * 16);
                bn[t][g] = *reinterpret_cast<const v4i *>(base + OPB + (kh * 4 + g) * 2048 + (64 * wn + 32 * t + r32) * 16);
            }
            constexpr int PI[10] = {0, 0, 1, 0, 1, 2, 0, 1, 2, 3};
            constexpr int PJ[10] = {0, 1, 0, 2, 1, 0, 3, 2, 1, 0};
            constexpr int LO[5] = {0, 3, 6, 8, 10};
#pragma unroll
            for (int q = LO[g]; q < LO[g + 1]; ++q)
#pragma unroll
                for (int ta = 0; ta < 2; ++ta)
#pragma unroll
                    for (int tb = 0; tb < 2; ++tb)
                        acc[ta][tb][PI[q] + PJ[q]] =
                            __builtin_amdgcn_mfma_i32_32x32x32_i8(a[ta][PI[q]], b[tb][PJ[q]], acc[ta][tb][PI[q] + PJ[q]], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
        }
    };
    v4i a0[2][4], b0[2][4], a1[2][4], b1[2][4];
    // nst is a multiple of 4 (Dp a multiple of 128)
#pragma unroll
    for (int q = 0; q < 4; ++q)
#pragma unroll
        for (int g = 0; g < 4; ++g) issue1(pcur + q * sstride, q, g);
    asm volatile("s_waitcnt vmcnt(24)" ::: "memory");
    __builtin_amdgcn_s_barrier();
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int l = 0; l < 4; ++l) {
            a0[t][l] = *reinterpret_cast<const v4i *>(smem + (kh * 4 + l) * 2048 + (64 * wm + 32 * t + r32) * 16);
            b0[t][l] = *reinterpret_cast<const v4i *>(smem + OPB + (kh * 4 + l) * 2048 + (64 * wn + 32 * t + r32) * 16);
        }
    for (long q = 0;; ++q) {
        const bool has_next = tile_of(q + 1, m0n, n0n);
        pnext = base_of(m0n, n0n);
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
            for (int b = 0; b < 2; ++b)
#pragma unroll
                for (int s = 0; s < 4; ++s)
#pragma unroll
                    for (int v = 0; v < 16; ++v) acc[a][b][s][v] = 0;
        // before the reads of stage st + 1: it has landed everywhere and every wave has left slot st % NBUF.  Two later
        // stages stay in flight (stores of the previous tile's results only make the count conservative)
        auto sync = [&](int st) {
            const int ahead = (mode & 16) ? 0 : has_next ? 2 : (nst - 1 < st + 3 ? nst - 1 : st + 3) - (st + 1);
            if (ahead >= 2) asm volatile("s_waitcnt vmcnt(16)\n\ts_waitcnt lgkmcnt(0)" ::: "memory");
            else if (ahead == 1) asm volatile("s_waitcnt vmcnt(8)\n\ts_waitcnt lgkmcnt(0)" ::: "memory");
            else asm volatile("s_waitcnt vmcnt(0)\n\ts_waitcnt lgkmcnt(0)" ::: "memory");
            if (!(mode & 64)) __builtin_amdgcn_s_barrier();
        };
        for (int st = 0; st < nst; st += 2) {
            sync(st);
            step(st, has_next, a0, b0, a1, b1);
            sync(st + 1);
            step(st + 1, has_next, a1, b1, a0, b0);      // (reads past the end of the last tile hit a slot nobody uses)
        }
#pragma unroll
        for (int ta = 0; ta < 2; ++ta)
#pragma unroll
            for (int tb = 0; tb < 2; ++tb)
#pragma unroll
                for (int v = 0; v < 16; ++v) {
                    if ((mode == 5 || (mode & 128)) && v != 0) continue;
                    const long row = m0 + 64 * wm + 32 * ta + 8 * (v >> 2) + 4 * kh + (v & 3);
                    const int col = n0 + 64 * wn + 32 * tb + r32;
                    float t = (float)acc[ta][tb][3][v];
                    t = __builtin_fmaf((float)acc[ta][tb][2][v], 256.0f, t);
                    t = __builtin_fmaf((float)acc[ta][tb][1][v], 65536.0f, t);
                    t = __builtin_fmaf((float)acc[ta][tb][0][v], 16777216.0f, t);
                    if (mode != 1 || t == 1.2345e30f) out[row * RB + col] = t;
                }
        if (!has_next) break;
        m0 = m0n;
        n0 = n0n;
        pcur = pnext;
    }
}

int main(int argc, char **argv) {
    const long RA = argc > 1 ? atol(argv[1]) : 65536;
    const long RB = argc > 2 ? atol(argv[2]) : 2048;
    const int Dp = argc > 3 ? atoi(argv[3]) : 512;
    const int mode = MODE;
    const size_t na = (size_t)RA * Dp * 4, nb = (size_t)RB * Dp * 4;
    std::vector<int8_t> ha(na), hb(nb);
    uint32_t s = 12345;
    auto rnd = [&]() { s = s * 1664525u + 1013904223u; return (int8_t)(s >> 24); };
    for (auto &v : ha) v = rnd();
    for (auto &v : hb) v = rnd();
    int8_t *da, *db;
    float *dout;
    CK(hipMalloc(&da, na));
    CK(hipMalloc(&db, nb));
    CK(hipMalloc(&dout, (size_t)RA * RB * 4));
    CK(hipMemcpy(da, ha.data(), na, hipMemcpyHostToDevice));
    CK(hipMemcpy(db, hb.data(), nb, hipMemcpyHostToDevice));
    const int lds = NBUF * SLOT;
    CK(hipFuncSetAttribute((const void *)k_limb_gemm, hipFuncAttributeMaxDynamicSharedMemorySize, lds));
    const long mts = (RA + TM - 1) / TM;
    const long tiles = ((mts + 7) / 8) * 8 * (RB / TN);
    const unsigned grid = (unsigned)(tiles < 256 ? tiles : 256);
    (void)mts;
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    for (int rep = 0; rep < 3; ++rep) hipLaunchKernelGGL(k_limb_gemm, dim3(grid), dim3(256), lds, 0, da, RA, db, RB, Dp, dout, mode);
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0));
    const int reps = 20;
    for (int rep = 0; rep < reps; ++rep) hipLaunchKernelGGL(k_limb_gemm, dim3(grid), dim3(256), lds, 0, da, RA, db, RB, Dp, dout, mode);
    CK(hipEventRecord(e1));
    CK(hipDeviceSynchronize());
    float ms;
    CK(hipEventElapsedTime(&ms, e0, e1));
    ms /= reps;
    const double ops = 2.0 * RA * RB * Dp * 10;
    printf("limb gemm %ld x %ld x %d: %.3f ms, %.1f TOPS int8, fp32-equivalent %.1f TFLOP/s\n", RA, RB, Dp, ms, ops / ms * 1e-9,
           2.0 * RA * RB * Dp / ms * 1e-9);
    // check a sample
    std::vector<float> ho(RB * 4);
    long bad = 0;
    const long rows[4] = {0, 77, RA / 2 + 5, RA - 1};
    for (int q = 0; q < 4; ++q) {
        CK(hipMemcpy(ho.data() + q * RB, dout + rows[q] * RB, RB * 4, hipMemcpyDeviceToHost));
        for (long c = 0; c < RB; ++c) {
            int64_t P[4] = {0, 0, 0, 0};
            for (int k = 0; k < Dp; ++k) {
                int64_t la[4], lb[4];
                for (int l = 0; l < 4; ++l) {
                    la[l] = ha[(((size_t)(k / 16) * 4 + l) * RA + rows[q]) * 16 + k % 16];
                    lb[l] = hb[(((size_t)(k / 16) * 4 + l) * RB + c) * 16 + k % 16];
                }
                for (int i = 0; i < 4; ++i)
                    for (int jj = 0; jj + i < 4; ++jj) P[i + jj] += la[i] * lb[jj];
            }
            float t = (float)(int32_t)P[3];
            t = fmaf((float)(int32_t)P[2], 256.0f, t);
            t = fmaf((float)(int32_t)P[1], 65536.0f, t);
            t = fmaf((float)(int32_t)P[0], 16777216.0f, t);
            if (ho[q * RB + c] != t) {
                if (bad < 5) printf("mismatch row %ld col %ld: %f vs %f\n", rows[q], c, ho[q * RB + c], t);
                ++bad;
            }
        }
    }
    printf("checked %ld outputs, %ld mismatches\n", 4 * RB, bad);
    return mode == 0 && bad != 0;
}
