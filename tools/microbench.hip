// microbench.hip — machine ceilings behind the PG-SGD kernel design (MI355X):
// random 16-byte gathers, random fp32 atomic adds (1, 2 adjacent, 4 = the kernel's pattern),
// 64-bit integer atomics, over working sets from L2-resident to HBM-resident.
// Build: hipcc -O3 --offload-arch=gfx950 tools/microbench.hip -o odgi_amd/lib/microbench
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
#include <vector>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

__device__ __forceinline__ uint64_t xs(uint64_t& s) {  // xorshift64*
    s ^= s >> 12; s ^= s << 25; s ^= s >> 27;
    return s * 0x2545F4914F6CDD1Dull;
}

template <int ILP>
__global__ void gather16(const uint4* buf, uint64_t n_elems, uint64_t iters, uint32_t* sink) {
    uint64_t s = 0x9E3779B97F4A7C15ull * (blockIdx.x * (uint64_t)blockDim.x + threadIdx.x + 1);
    uint32_t acc = 0;
    for (uint64_t i = 0; i < iters; ++i) {
        uint4 v[ILP];
#pragma unroll
        for (int j = 0; j < ILP; ++j) v[j] = buf[__umul64hi(xs(s), n_elems)];
#pragma unroll
        for (int j = 0; j < ILP; ++j) acc += v[j].x ^ v[j].w;
    }
    if (acc == 0x12345678u) *sink = acc;
}

// dependent pair: second gather lands near the first (the Zipf partner pattern)
__global__ void gather16_near(const uint4* buf, uint64_t n_elems, uint64_t iters, uint32_t* sink) {
    uint64_t s = 0x9E3779B97F4A7C15ull * (blockIdx.x * (uint64_t)blockDim.x + threadIdx.x + 1);
    uint32_t acc = 0;
    for (uint64_t i = 0; i < iters; ++i) {
        const uint64_t k = __umul64hi(xs(s), n_elems - 64);
        const uint4 a = buf[k];
        const uint4 b = buf[k + 1 + (xs(s) & 31)];
        acc += a.x ^ b.w;
    }
    if (acc == 0x12345678u) *sink = acc;
}

// MODE 0: one f32 atomic; 1: two adjacent (x,y); 2: four (two ends, the kernel's pattern);
// 3: one u64 integer atomic; 4: two u64 integer atomics (packed x|y per end); 5: one f64 atomic
template <int MODE>
__global__ void atomics(float* buf, uint64_t n_slots8, uint64_t iters) {
    uint64_t s = 0x9E3779B97F4A7C15ull * (blockIdx.x * (uint64_t)blockDim.x + threadIdx.x + 1);
    for (uint64_t i = 0; i < iters; ++i) {
        const uint64_t a = __umul64hi(xs(s), n_slots8), b = __umul64hi(xs(s), n_slots8);
        float* pa = buf + 2 * a;
        float* pb = buf + 2 * b;
        if (MODE == 0) { unsafeAtomicAdd(pa, 1.0f); }
        if (MODE == 1) { unsafeAtomicAdd(pa, 1.0f); unsafeAtomicAdd(pa + 1, 1.0f); }
        if (MODE == 2) { unsafeAtomicAdd(pa, 1.0f); unsafeAtomicAdd(pa + 1, 1.0f); unsafeAtomicAdd(pb, -1.0f); unsafeAtomicAdd(pb + 1, -1.0f); }
        if (MODE == 3) { atomicAdd((unsigned long long*)pa, 0x100000001ull); }
        if (MODE == 4) { atomicAdd((unsigned long long*)pa, 0x100000001ull); atomicAdd((unsigned long long*)pb, 0xffffffffffffffffull); }
        if (MODE == 5) { unsafeAtomicAdd((double*)pa, 1.0); }
    }
}

// MODE 0: two plain 8-byte stores; 1: two agent-scope (sc1, write-through) 8-byte stores;
// 2: two 64-bit CAS with the expected value just loaded (sc1 load + CAS per end);
// 3: two sc1 8-byte loads + two sc1 8-byte stores (the Hogwild load-then-store pattern)
template <int MODE>
__global__ void stores(float* buf, uint64_t n_slots8, uint64_t iters) {
    uint64_t s = 0x9E3779B97F4A7C15ull * (blockIdx.x * (uint64_t)blockDim.x + threadIdx.x + 1);
    for (uint64_t i = 0; i < iters; ++i) {
        const uint64_t a = __umul64hi(xs(s), n_slots8), b = __umul64hi(xs(s), n_slots8);
        uint64_t* pa = (uint64_t*)(buf + 2 * a);
        uint64_t* pb = (uint64_t*)(buf + 2 * b);
        if (MODE == 0) { *pa = s; *pb = s + 1; }
        if (MODE == 1) { __hip_atomic_store(pa, s, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); __hip_atomic_store(pb, s + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
        if (MODE == 2) {
            uint64_t ea = __hip_atomic_load(pa, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            uint64_t eb = __hip_atomic_load(pb, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_compare_exchange_strong(pa, &ea, ea + 1, __ATOMIC_RELAXED, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_compare_exchange_strong(pb, &eb, eb + 1, __ATOMIC_RELAXED, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        if (MODE == 3) {
            const uint64_t ea = __hip_atomic_load(pa, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            const uint64_t eb = __hip_atomic_load(pb, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(pa, ea + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(pb, eb + 3, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
}

// loads of the coordinate pattern: two 8-byte agent-scope loads per iteration
template <bool SC1>
__global__ void loads8(const float* buf, uint64_t n_slots8, uint64_t iters, uint32_t* sink) {
    uint64_t s = 0x9E3779B97F4A7C15ull * (blockIdx.x * (uint64_t)blockDim.x + threadIdx.x + 1);
    uint64_t acc = 0;
    for (uint64_t i = 0; i < iters; ++i) {
        const uint64_t a = __umul64hi(xs(s), n_slots8), b = __umul64hi(xs(s), n_slots8);
        if (SC1) {
            acc += __hip_atomic_load((const uint64_t*)(buf + 2 * a), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            acc += __hip_atomic_load((const uint64_t*)(buf + 2 * b), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        } else {
            acc += *(const uint64_t*)(buf + 2 * a) + *(const uint64_t*)(buf + 2 * b);
        }
    }
    if (acc == 0x12345678u) *sink = (uint32_t)acc;
}

template <typename F>
static double time_ms(F launch, int reps = 3) {
    hipEvent_t a, b;
    hipEventCreate(&a); hipEventCreate(&b);
    launch();
    hipDeviceSynchronize();
    float best = 1e30f;
    for (int r = 0; r < reps; ++r) {
        hipEventRecord(a); launch(); hipEventRecord(b); hipEventSynchronize(b);
        float ms; hipEventElapsedTime(&ms, a, b);
        if (ms < best) best = ms;
    }
    hipEventDestroy(a); hipEventDestroy(b);
    return best;
}

int main() {
    const int grid = 256 * 8, block = 256;
    const uint64_t lanes = (uint64_t)grid * block;
    uint32_t* sink; CK(hipMalloc(&sink, 4));
    const size_t big = 1ull << 30;  // 1 GiB
    uint4* g; CK(hipMalloc(&g, big)); CK(hipMemset(g, 1, big));
    printf("{\"lanes\": %llu}\n", (unsigned long long)lanes);
    for (uint64_t bytes : {16ull << 20, 128ull << 20, 768ull << 20}) {
        const uint64_t n = bytes / 16, iters = 256;
        double ms1 = time_ms([&] { hipLaunchKernelGGL(gather16<1>, dim3(grid), dim3(block), 0, 0, g, n, iters, sink); });
        double ms2 = time_ms([&] { hipLaunchKernelGGL(gather16<2>, dim3(grid), dim3(block), 0, 0, g, n, iters / 2, sink); });
        double ms4 = time_ms([&] { hipLaunchKernelGGL(gather16<4>, dim3(grid), dim3(block), 0, 0, g, n, iters / 4, sink); });
        double msn = time_ms([&] { hipLaunchKernelGGL(gather16_near, dim3(grid), dim3(block), 0, 0, g, n, iters / 2, sink); });
        const double total = (double)lanes * iters;
        printf("{\"bench\": \"gather16\", \"MiB\": %llu, \"G_per_s_ilp1\": %.2f, \"ilp2\": %.2f, \"ilp4\": %.2f, \"near_pairs_G_gathers_per_s\": %.2f}\n",
               (unsigned long long)(bytes >> 20), total / ms1 / 1e6, total / ms2 / 1e6, total / ms4 / 1e6, total / msn / 1e6);
    }
    float* c = (float*)g;
    for (uint64_t bytes : {1ull << 20, 16ull << 20, 160ull << 20, 768ull << 20}) {
        const uint64_t n8 = bytes / 8, iters = 128;
        const double total = (double)lanes * iters;
        double m0 = time_ms([&] { hipLaunchKernelGGL(atomics<0>, dim3(grid), dim3(block), 0, 0, c, n8, iters); });
        double m1 = time_ms([&] { hipLaunchKernelGGL(atomics<1>, dim3(grid), dim3(block), 0, 0, c, n8, iters); });
        double m2 = time_ms([&] { hipLaunchKernelGGL(atomics<2>, dim3(grid), dim3(block), 0, 0, c, n8, iters); });
        double m3 = time_ms([&] { hipLaunchKernelGGL(atomics<3>, dim3(grid), dim3(block), 0, 0, c, n8, iters); });
        double m4 = time_ms([&] { hipLaunchKernelGGL(atomics<4>, dim3(grid), dim3(block), 0, 0, c, n8, iters); });
        double m5 = time_ms([&] { hipLaunchKernelGGL(atomics<5>, dim3(grid), dim3(block), 0, 0, c, n8, iters); });
        double l0 = time_ms([&] { hipLaunchKernelGGL(loads8<false>, dim3(grid), dim3(block), 0, 0, c, n8, iters, sink); });
        double l1 = time_ms([&] { hipLaunchKernelGGL(loads8<true>, dim3(grid), dim3(block), 0, 0, c, n8, iters, sink); });
        double s0 = time_ms([&] { hipLaunchKernelGGL(stores<0>, dim3(grid), dim3(block), 0, 0, c, n8, iters); });
        double s1 = time_ms([&] { hipLaunchKernelGGL(stores<1>, dim3(grid), dim3(block), 0, 0, c, n8, iters); });
        double s2 = time_ms([&] { hipLaunchKernelGGL(stores<2>, dim3(grid), dim3(block), 0, 0, c, n8, iters); });
        double s3 = time_ms([&] { hipLaunchKernelGGL(stores<3>, dim3(grid), dim3(block), 0, 0, c, n8, iters); });
        printf("{\"bench\": \"stores\", \"MiB\": %llu, \"Giter_per_s\": {\"store8x2_plain\": %.2f, \"store8x2_sc1\": %.2f, \"load+cas64_x2\": %.2f, \"load_sc1+store_sc1_x2\": %.2f}}\n",
               (unsigned long long)(bytes >> 20), total / s0 / 1e6, total / s1 / 1e6, total / s2 / 1e6, total / s3 / 1e6);
        printf("{\"bench\": \"atomics\", \"MiB\": %llu, \"Giter_per_s\": {\"f32x1\": %.2f, \"f32x2_adjacent\": %.2f, \"f32x4_two_ends\": %.2f, "
               "\"u64x1\": %.2f, \"u64x2_two_ends\": %.2f, \"f64x1\": %.2f, \"load8x2_plain\": %.2f, \"load8x2_sc1\": %.2f}}\n",
               (unsigned long long)(bytes >> 20), total / m0 / 1e6, total / m1 / 1e6, total / m2 / 1e6, total / m3 / 1e6, total / m4 / 1e6,
               total / m5 / 1e6, total / l0 / 1e6, total / l1 / 1e6);
    }
    return 0;
}
