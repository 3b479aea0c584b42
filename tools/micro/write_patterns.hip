// write_patterns.hip -- what the OUTPUT side of decode costs by itself: B rows of R bytes written (a) as a plain contiguous
// fill, (b) as decode's LDS-resident kernels write them: a persistent workgroup per CU owns a P-byte slice of every row and
// walks a contiguous range of rows, P = 64 ... 2048, XCD x owning a run of adjacent slices.  Burst (cold) and sustained
// (50 launches after 50 untimed ones) times; nontemporal and plain stores.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <bool NT> __device__ inline void st(f32x4 *p, f32x4 v) {
    if (NT) __builtin_nontemporal_store(v, p); else *p = v;
}

template <bool NT>
__global__ void __launch_bounds__(1024) k_fill(f32x4 *__restrict__ out, long n16) {
    const f32x4 v = (f32x4){1.f, 2.f, 3.f, 4.f};
    for (long i = (long)blockIdx.x * 1024 + threadIdx.x; i < n16; i += (long)gridDim.x * 1024) st<NT>(out + i, v);
}

// contiguous fill, every workgroup a contiguous range (chunked)
template <bool NT>
__global__ void __launch_bounds__(1024) k_fill_chunk(f32x4 *__restrict__ out, long n16) {
    const f32x4 v = (f32x4){1.f, 2.f, 3.f, 4.f};
    const long per = (n16 + gridDim.x - 1) / gridDim.x;
    const long lo = blockIdx.x * per, hi = lo + per < n16 ? lo + per : n16;
    for (long i = lo + threadIdx.x; i < hi; i += 1024) st<NT>(out + i, v);
}

// decode's pattern: slices of P bytes, LPV = P / 16 lanes per (row, slice)
template <int P, bool NT, int UNR>
__global__ void __launch_bounds__(1024) k_slice(float *__restrict__ out, long B, int R /* floats per row */, int groups) {
    constexpr int LPV = P / 16;
    const int ns = R * 4 / P;
    const int xcd = blockIdx.x & 7, within = blockIdx.x >> 3;
    const int per_xcd = (ns + 7) / 8;
    const int slice = xcd * per_xcd + within % per_xcd;
    const int grp = within / per_xcd;
    if (slice >= ns) return;
    const long per = (B + groups - 1) / groups;
    const long b_lo = grp * per, b_hi = (b_lo + per < B) ? b_lo + per : B;
    const int q = threadIdx.x % LPV;
    const long stride = 1024 / LPV;
    const int off = slice * (P / 4) + 4 * q;
    f32x4 v = (f32x4){1.f, 2.f, 3.f, (float)slice};
    for (long b = b_lo + threadIdx.x / LPV; b < b_hi; b += UNR * stride) {
#pragma unroll
        for (int u = 0; u < UNR; ++u)
            if (b + u * stride < b_hi) st<NT>(reinterpret_cast<f32x4 *>(out + (b + u * stride) * R + off), v);
    }
}

// the same 64-byte slices, but the workgroups of a slice interleave their rows (row b goes to group b mod groups): every
// workgroup's write front moves through the whole buffer
template <bool NT>
__global__ void __launch_bounds__(1024) k_slice_il(float *__restrict__ out, long B, int R, int groups) {
    constexpr int LPV = 4;
    const int ns = R / 16;
    const int xcd = blockIdx.x & 7, within = blockIdx.x >> 3;
    const int per_xcd = (ns + 7) / 8;
    const int slice = xcd * per_xcd + within % per_xcd;
    const int grp = within / per_xcd;
    if (slice >= ns) return;
    const int q = threadIdx.x % LPV;
    const int off = slice * 16 + 4 * q;
    f32x4 v = (f32x4){1.f, 2.f, 3.f, (float)slice};
    // blocks of 256 consecutive rows dealt round robin to the groups
    for (long blk = grp; blk * 256 < B; blk += groups) {
        const long b = blk * 256 + threadIdx.x / LPV;
        if (b < B) st<NT>(reinterpret_cast<f32x4 *>(out + b * R + off), v);
    }
}

static hipEvent_t e0, e1;
template <typename F> void timeit(const char *name, size_t bytes, F launch) {
    // burst: cold-ish (after a big unrelated fill), 20 launches
    launch(); hipDeviceSynchronize();
    hipEventRecord(e0);
    for (int r = 0; r < 20; ++r) launch();
    hipEventRecord(e1); hipDeviceSynchronize();
    float ms20; hipEventElapsedTime(&ms20, e0, e1);
    for (int r = 0; r < 50; ++r) launch();
    hipEventRecord(e0);
    for (int r = 0; r < 50; ++r) launch();
    hipEventRecord(e1); hipDeviceSynchronize();
    float ms50; hipEventElapsedTime(&ms50, e0, e1);
    printf("%-44s first 20: %7.1f us (%5.2f TB/s)   sustained: %7.1f us (%5.2f TB/s)\n", name, ms20 * 50.f, bytes / (ms20 / 20 * 1e-3) / 1e12,
           ms50 * 20.f, bytes / (ms50 / 50 * 1e-3) / 1e12);
}

template <int P, bool NT, int UNR> void run_slice(float *out, long B, int R) {
    const int ns = R * 4 / P, per_xcd = (ns + 7) / 8;
    int groups = 256 / (8 * per_xcd); if (groups < 1) groups = 1;
    const unsigned g = 8 * per_xcd * groups;
    char name[96]; snprintf(name, sizeof name, "slices of %4d B, %s, %d per trip, grid %u", P, NT ? "nontemporal" : "plain", UNR, g);
    timeit(name, (size_t)B * R * 4, [&] { hipLaunchKernelGGL((k_slice<P, NT, UNR>), dim3(g), dim3(1024), 0, 0, out, B, R, groups); });
}

int main(int argc, char **argv) {
    hipEventCreate(&e0); hipEventCreate(&e1);
    const int R = argc > 2 ? atoi(argv[2]) : 512;
    for (long B : {65536L, 1048576L}) {
        if (argc > 1 && atol(argv[1]) > 0 && B != atol(argv[1])) continue;
        const size_t bytes = (size_t)B * R * 4;
        float *out; hipMalloc(&out, bytes);
        printf("--- %ld rows of %d bytes (%.0f MB)\n", B, R * 4, bytes / 1e6);
        const long n16 = bytes / 16;
        timeit("fill, grid-stride, 256 x 1024, nontemporal", bytes, [&] { hipLaunchKernelGGL((k_fill<true>), dim3(256), dim3(1024), 0, 0, (f32x4 *)out, n16); });
        timeit("fill, grid-stride, 256 x 1024, plain", bytes, [&] { hipLaunchKernelGGL((k_fill<false>), dim3(256), dim3(1024), 0, 0, (f32x4 *)out, n16); });
        timeit("fill, grid-stride, 2048 x 1024, plain", bytes, [&] { hipLaunchKernelGGL((k_fill<false>), dim3(2048), dim3(1024), 0, 0, (f32x4 *)out, n16); });
        timeit("fill, one float4 per thread, plain", bytes, [&] { hipLaunchKernelGGL((k_fill<false>), dim3((unsigned)(n16 / 1024)), dim3(1024), 0, 0, (f32x4 *)out, n16); });
        timeit("fill, contiguous range per WG, 256, nt", bytes, [&] { hipLaunchKernelGGL((k_fill_chunk<true>), dim3(256), dim3(1024), 0, 0, (f32x4 *)out, n16); });
        timeit("hipMemsetAsync", bytes, [&] { hipMemsetAsync(out, 0, bytes, 0); });
        run_slice<64, true, 1>(out, B, R);
        run_slice<64, false, 1>(out, B, R);
        run_slice<64, true, 2>(out, B, R);
        run_slice<64, true, 4>(out, B, R);
        run_slice<128, true, 1>(out, B, R);
        run_slice<128, false, 1>(out, B, R);
        run_slice<256, true, 1>(out, B, R);
        run_slice<256, false, 1>(out, B, R);
        run_slice<512, true, 1>(out, B, R);
        run_slice<1024, true, 1>(out, B, R);
        run_slice<2048, true, 1>(out, B, R);
        {
            const int ns = R / 16, per_xcd = (ns + 7) / 8; int groups = 256 / (8 * per_xcd); if (groups < 1) groups = 1;
            const unsigned g = 8 * per_xcd * groups;
            timeit("slices of 64 B, rows dealt in blocks of 256, nt", bytes, [&] { hipLaunchKernelGGL((k_slice_il<true>), dim3(g), dim3(1024), 0, 0, out, B, R, groups); });
            timeit("slices of 64 B, rows dealt in blocks, plain", bytes, [&] { hipLaunchKernelGGL((k_slice_il<false>), dim3(g), dim3(1024), 0, 0, out, B, R, groups); });
        }
        hipFree(out);
    }
    return 0;
}
