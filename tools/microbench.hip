// microbench.hip — machine ceilings behind the PG-SGD kernel design (MI355X):
// random 16-byte gathers, random fp32 atomic adds (1, 2 adjacent, 4 = the kernel's pattern),
// 64-bit integer atomics, over working sets from L2-resident to HBM-resident.
// Build: hipcc -O3 --offload-arch=gfx950 tools/microbench.hip -o odgi_amd/lib/microbench
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
#include <cstdlib>
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

// round 2 -----------------------------------------------------------------------------------------------------
// 32-byte gathers (the tile kernel's far-partner record with the coordinate snapshot inside)
__global__ void gather32(const uint4* buf, uint64_t n_elems32, uint64_t iters, uint32_t* sink) {
    uint64_t s = 0x9E3779B97F4A7C15ull * (blockIdx.x * (uint64_t)blockDim.x + threadIdx.x + 1);
    uint32_t acc = 0;
    for (uint64_t i = 0; i < iters; ++i) {
        const uint64_t k = __umul64hi(xs(s), n_elems32);
        const uint4 a = buf[2 * k], b = buf[2 * k + 1];
        acc += a.x ^ b.w;
    }
    if (acc == 0x12345678u) *sink = acc;
}

// 64-bit atomic adds at workgroup scope (performed in the issuing XCD's L2) against agent scope
template <int SCOPE>
__global__ void atomics_scope(unsigned long long* buf, uint64_t n_slots, uint64_t iters) {
    uint64_t s = 0x9E3779B97F4A7C15ull * (blockIdx.x * (uint64_t)blockDim.x + threadIdx.x + 1);
    for (uint64_t i = 0; i < iters; ++i) {
        const uint64_t a = __umul64hi(xs(s), n_slots);
        __hip_atomic_fetch_add(buf + a, 0x100000001ull, __ATOMIC_RELAXED, SCOPE);
    }
}

// The far-update outbox: every lane appends 16-byte messages to one of B destination buckets.  A workgroup owns, per
// bucket, a private chunk of CH messages in global memory; the slot inside the chunk comes from an LDS atomic on a
// packed (chunk id, fill) word, a new chunk from one returning global atomic on the bucket's chunk counter.
// SKEW: 0 = destinations uniform over the buckets; 1 = half of them in the workgroup's own bucket +-1.
template <int CH, int SKEW>
__global__ void outbox(uint4* pool, unsigned int* bucket_next, uint32_t B, uint32_t cap_chunks, uint64_t iters, unsigned int* overflow) {
    extern __shared__ unsigned long long st[];
    for (uint32_t b = threadIdx.x; b < B; b += blockDim.x) st[b] = (0xffffffffull << 32) | (unsigned)CH;
    __syncthreads();
    uint64_t s = 0x9E3779B97F4A7C15ull * (blockIdx.x * (uint64_t)blockDim.x + threadIdx.x + 1);
    const uint32_t home = blockIdx.x % B;
    for (uint64_t i = 0; i < iters; ++i) {
        const uint64_t r = xs(s);
        uint32_t b = (uint32_t)__umul64hi(r, B);
        if (SKEW && (r & 1)) b = (home + B + (uint32_t)((r >> 1) % 3) - 1) % B;
        const uint4 msg = make_uint4((uint32_t)r, (uint32_t)(r >> 32), (uint32_t)i, b);
        for (;;) {
            const unsigned long long old = atomicAdd(&st[b], 1ull);
            const uint32_t slot = (uint32_t)old, chunk = (uint32_t)(old >> 32);
            if (slot < (uint32_t)CH) {
                pool[((uint64_t)b * cap_chunks + chunk) * CH + slot] = msg;
                break;
            }
            if (slot == (uint32_t)CH) {
                uint32_t nc = atomicAdd(bucket_next + b, 1u);
                if (nc >= cap_chunks) { atomicAdd(overflow, 1u); nc = cap_chunks - 1; }
                atomicExch(&st[b], (unsigned long long)nc << 32);
            } else {
                __builtin_amdgcn_s_sleep(1);
            }
        }
    }
}

// the same messages sent as one agent-scope 64-bit atomic each (what the tile kernel does today)
__global__ void outbox_atomic(unsigned long long* coords, uint64_t n_slots, uint64_t iters) {
    uint64_t s = 0x9E3779B97F4A7C15ull * (blockIdx.x * (uint64_t)blockDim.x + threadIdx.x + 1);
    for (uint64_t i = 0; i < iters; ++i) atomicAdd(coords + __umul64hi(xs(s), n_slots), 0x100000001ull);
}

// drain: one workgroup per bucket streams the bucket's chunks and adds the messages into an LDS window
template <int CH>
__global__ void drain(const uint4* pool, const unsigned int* bucket_next, uint32_t cap_chunks, unsigned long long* coords, uint32_t ends_per_bucket) {
    extern __shared__ unsigned long long acc[];
    const uint32_t b = blockIdx.x;
    for (uint32_t i = threadIdx.x; i < ends_per_bucket; i += blockDim.x) acc[i] = 0;
    __syncthreads();
    const uint64_t n_msgs = (uint64_t)min(bucket_next[b], cap_chunks) * CH;
    const uint4* src = pool + (uint64_t)b * cap_chunks * CH;
    for (uint64_t i = threadIdx.x; i < n_msgs; i += blockDim.x) {
        const uint4 m = src[i];
        atomicAdd(&acc[m.x % ends_per_bucket], (unsigned long long)m.y | ((unsigned long long)m.z << 32));
    }
    __syncthreads();
    for (uint32_t i = threadIdx.x; i < ends_per_bucket; i += blockDim.x) coords[(uint64_t)b * ends_per_bucket + i] += acc[i];
}

// Scattered writes of G adjacent 16-byte pieces (G lanes of one wave, one instruction) to random G*16-byte aligned
// places of a buffer far larger than the caches: does a write that covers a whole 64- or 128-byte unit avoid the
// read-for-ownership a 16-byte one pays?
template <int G>
__global__ void wstore(uint4* buf, uint64_t n_groups, uint64_t iters) {
    const uint32_t lane = threadIdx.x & 63u, grp = lane / G, piece = lane % G;
    uint64_t s = 0x9E3779B97F4A7C15ull * (((blockIdx.x * (uint64_t)blockDim.x + threadIdx.x) / G) * G + 1);  // same stream inside a group
    (void)grp;
    for (uint64_t i = 0; i < iters; ++i) {
        const uint64_t r = xs(s);
        buf[__umul64hi(r, n_groups) * G + piece] = make_uint4((uint32_t)r, (uint32_t)i, piece, 0u);
    }
}

// streaming writes: MODE 0 = 16 bytes into every 32-byte record (what snapshot_kernel does), 1 = the whole 32-byte
// record, 2 = a dense 16-byte array
template <int MODE>
__global__ void swrite(uint4* buf, uint64_t n, const uint32_t* src) {
    for (uint64_t k = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; k < n; k += (uint64_t)gridDim.x * blockDim.x) {
        const uint32_t h = src[k];
        const uint4 v = make_uint4(h, h + 1, h + 2, h + 3);
        if (MODE == 0) buf[2 * k + 1] = v;
        if (MODE == 1) { buf[2 * k] = v; buf[2 * k + 1] = v; }
        if (MODE == 2) buf[k] = v;
    }
}

// round 4: kernels of KNOWN traffic for calibrating rocprofv3's memory-side counters (FETCH_SIZE, WRITE_SIZE,
// TCC_EA0_RDREQ / _RDREQ_32B, TCC_EA0_WRREQ / _WRREQ_64B, TCC_HIT / TCC_MISS) on the access patterns of the tile kernel:
// MICROBENCH_CAL=1 launches each once (plus a warm-up of another name) and prints what it requested.
__global__ void cal_stream_read(const uint4* buf, uint64_t n16, uint32_t* sink) {   // coalesced 16 bytes per lane
    uint32_t acc = 0;
    for (uint64_t k = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; k < n16; k += (uint64_t)gridDim.x * blockDim.x) acc += buf[k].x;
    if (acc == 0x12345678u) *sink = acc;
}
__global__ void cal_stream_write(uint4* buf, uint64_t n16) {
    for (uint64_t k = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; k < n16; k += (uint64_t)gridDim.x * blockDim.x) buf[k] = make_uint4((uint32_t)k, 1u, 2u, 3u);
}
__global__ void cal_gather16(const uint4* buf, uint64_t n16, uint64_t iters, uint32_t* sink) {   // random 16-byte records
    uint64_t s = 0x9E3779B97F4A7C15ull * (blockIdx.x * (uint64_t)blockDim.x + threadIdx.x + 1);
    uint32_t acc = 0;
    for (uint64_t i = 0; i < iters; ++i) acc += buf[__umul64hi(xs(s), n16)].x;
    if (acc == 0x12345678u) *sink = acc;
}
__global__ void cal_gather32(const uint4* buf, uint64_t n32, uint64_t iters, uint32_t* sink) {   // the tile kernel's far partner: 16 + 8 bytes of a 32-byte record
    uint64_t s = 0x9E3779B97F4A7C15ull * (blockIdx.x * (uint64_t)blockDim.x + threadIdx.x + 1);
    uint32_t acc = 0;
    for (uint64_t i = 0; i < iters; ++i) {
        const uint64_t r = xs(s), k = __umul64hi(r, n32);
        const uint4 a = buf[2 * k];
        const unsigned long long w = reinterpret_cast<const unsigned long long*>(buf)[4 * k + 2 + (r & 1)];
        acc += a.x ^ (uint32_t)w;
    }
    if (acc == 0x12345678u) *sink = acc;
}
__global__ void cal_linewrite64(uint4* buf, uint64_t n_lines, uint64_t iters) {   // the outbox: 8 lanes x 8 bytes = one 64-byte line at a random place
    uint64_t s = 0x9E3779B97F4A7C15ull * (((blockIdx.x * (uint64_t)blockDim.x + threadIdx.x) / 8) * 8 + 1);
    const uint32_t piece = threadIdx.x & 7u;
    for (uint64_t i = 0; i < iters; ++i) {
        const uint64_t r = xs(s);
        reinterpret_cast<unsigned long long*>(buf)[__umul64hi(r, n_lines) * 8 + piece] = r + piece;
    }
}

// both 64-byte halves of one random 128-byte line, one after the other: one memory-side request per pair if a miss
// fetches the whole line, two if the L2 fetches 64-byte sectors
__global__ void cal_gather_halves(const uint4* buf, uint64_t n128, uint64_t iters, uint32_t* sink) {
    uint64_t s = 0x9E3779B97F4A7C15ull * (blockIdx.x * (uint64_t)blockDim.x + threadIdx.x + 1);
    uint32_t acc = 0;
    for (uint64_t i = 0; i < iters; ++i) {
        const uint64_t k = __umul64hi(xs(s), n128);
        const uint4 a = buf[8 * k];
        const uint4 b = buf[8 * k + 4 + (a.x & 1u)];   // (dependent: issued after the first half is back)
        acc += a.x ^ b.w;
    }
    if (acc == 0x12345678u) *sink = acc;
}
// 16-byte random gathers with cache-policy bits: MODE 0 plain, 1 nt, 2 sc1 (agent scope), 3 sc0 sc1 (system scope), 4 sc0 sc1 nt
template <int MODE>
__global__ void cal_gather16_policy(const uint4* buf, uint64_t n16, uint64_t iters, uint32_t* sink) {
    uint64_t s = 0x9E3779B97F4A7C15ull * (blockIdx.x * (uint64_t)blockDim.x + threadIdx.x + 1);
    uint32_t acc = 0;
    for (uint64_t i = 0; i < iters; ++i) {
        const uint4* p = buf + __umul64hi(xs(s), n16);
        uint4 v;
        if (MODE == 0) asm volatile("global_load_dwordx4 %0, %1, off\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
        if (MODE == 1) asm volatile("global_load_dwordx4 %0, %1, off nt\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
        if (MODE == 2) asm volatile("global_load_dwordx4 %0, %1, off sc1\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
        if (MODE == 3) asm volatile("global_load_dwordx4 %0, %1, off sc0 sc1\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
        if (MODE == 4) asm volatile("global_load_dwordx4 %0, %1, off sc0 sc1 nt\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
        acc += v.x;
    }
    if (acc == 0x12345678u) *sink = acc;
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

// MICROBENCH_LOCAL=1: does the random-request ceiling (~50 G/s) depend on where a wave's 64 requests go?  Every wave-iteration
// picks one aligned block of `block_bytes` at random (wave-uniform) and every lane one 64-byte sector in it (DISTINCT sectors
// when the block has exactly 64: a 4-KiB block is then read whole).
__global__ void cal_gather_wave_local(const uint4* buf, uint64_t bytes, uint64_t block_bytes, uint64_t iters, uint32_t* sink) {
    const uint64_t wave = (blockIdx.x * (uint64_t)blockDim.x + threadIdx.x) >> 6;
    uint64_t sw = 0xD1342543DE82EF95ull * (wave + 1);                                              // the wave's stream (same in all its lanes)
    uint64_t s = 0x9E3779B97F4A7C15ull * (blockIdx.x * (uint64_t)blockDim.x + threadIdx.x + 1);   // the lane's
    const uint64_t n_blocks = bytes / block_bytes, sectors = block_bytes / 64;
    uint32_t acc = 0;
    for (uint64_t i = 0; i < iters; ++i) {
        const uint64_t rw = xs(sw), b = __umul64hi(rw, n_blocks);
        const uint64_t sec = sectors == 64 ? ((threadIdx.x ^ rw) & 63u) : __umul64hi(xs(s), sectors);
        acc += buf[(b * block_bytes + sec * 64) / 16].x;
    }
    if (acc == 0x12345678u) *sink = acc;
}
// groups of G lanes share one random aligned run of G 32-byte records (G = 2: a 64-byte sector, the tile kernel's partner pairs; G = 4: a
// 128-byte line); every lane reads 16 + 8 bytes of its own record
template <int G>
__global__ void cal_gather_group(const uint4* buf, uint64_t n32, uint64_t iters, uint32_t* sink) {
    const uint64_t lane = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x;
    uint64_t s = 0x9E3779B97F4A7C15ull * (lane / G + 1);   // the group's stream
    uint32_t acc = 0;
    for (uint64_t i = 0; i < iters; ++i) {
        const uint64_t r = xs(s), k = __umul64hi(r, n32 / G) * G + lane % G;
        const uint4 a = buf[2 * k];
        const unsigned long long w = reinterpret_cast<const unsigned long long*>(buf)[4 * k + 2 + (r & 1)];
        acc += a.x ^ (uint32_t)w;
    }
    if (acc == 0x12345678u) *sink = acc;
}
// Does a gather fill a 64-byte sector or the whole 128-byte line?  Few lanes (the L2 is not turned over between a lane's two reads): a
// random record, then MODE 0 nothing, 1 the other 64-byte half of the same 128-byte line, 2 another random record (each read depends on the one before).
template <int MODE>
__global__ void cal_line_fill(const uint4* buf, uint64_t n128, uint64_t iters, uint32_t* sink) {
    uint64_t s = 0x9E3779B97F4A7C15ull * (blockIdx.x * (uint64_t)blockDim.x + threadIdx.x + 1);
    uint32_t acc = 0;
    for (uint64_t i = 0; i < iters; ++i) {
        const uint64_t r = xs(s) + acc, line = __umul64hi(r, n128), half = r & 1;
        const uint4 a = buf[line * 8 + half * 4];
        acc += a.x;
        if (MODE == 1) acc += buf[line * 8 + (half ^ 1) * 4 + (a.x & 3)].y;
        if (MODE == 2) acc += buf[__umul64hi(xs(s) + a.x, n128) * 8 + (a.x & 3)].y;
    }
    if (acc == 0x12345678u) *sink = acc;
}
int main() {
    const int grid = 256 * 8, block = 256;
    const uint64_t lanes = (uint64_t)grid * block;
    uint32_t* sink; CK(hipMalloc(&sink, 4));
    const size_t big = 1ull << 30;  // 1 GiB
    uint4* g; CK(hipMalloc(&g, big)); CK(hipMemset(g, 1, big));
    printf("{\"lanes\": %llu}\n", (unsigned long long)lanes);
    if (getenv("MICROBENCH_WS")) {   // the random-line ceiling against the working set (TLB reach): 32-byte records gathered from 1.5 ... 48 GB
        (void)hipFree(g);
        const uint64_t iters = 128;
        const double req = (double)lanes * iters;
        for (unsigned long long gb10 : {15ull, 60ull, 240ull, 480ull}) {
            const size_t bytes = (size_t)(gb10 * 100) << 20;
            uint4* w = nullptr;
            if (hipMalloc(&w, bytes) != hipSuccess) { printf("{\"ws\": \"allocation of %llu MB failed\"}\n", gb10 * 100); continue; }
            hipMemset(w, 1, bytes);
            for (int rep = 0; rep < 2; ++rep) {
                hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
                hipDeviceSynchronize();
                hipEventRecord(a);
                hipLaunchKernelGGL(cal_gather_group<1>, dim3(grid), dim3(block), 0, 0, w, (uint64_t)bytes / 32, iters, sink);
                hipEventRecord(b); hipEventSynchronize(b);
                float ms; hipEventElapsedTime(&ms, a, b);
                printf("{\"ws\": \"random 32-byte records\", \"working_set_MB\": %llu, \"ms\": %.4f, \"G_gathers_per_s\": %.2f}\n", gb10 * 100, ms, req / ms / 1e6);
                hipEventDestroy(a); hipEventDestroy(b);
            }
            (void)hipFree(w);
        }
        return 0;
    }
    if (getenv("MICROBENCH_LOCAL")) {
        (void)hipFree(g);
        const size_t big2 = 1536ull << 20;
        CK(hipMalloc(&g, big2)); CK(hipMemset(g, 1, big2));
        const uint64_t iters = 128;
        const double req = (double)lanes * iters;
        hipLaunchKernelGGL(cal_stream_read, dim3(4096), dim3(256), 0, 0, g, big2 / 16, sink);   // first touch
        for (int rep = 0; rep < 2; ++rep)
            for (unsigned long long bb : {4096ull, 16384ull, 65536ull, 1ull << 20, 16ull << 20, (unsigned long long)big2}) {
                hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
                hipDeviceSynchronize();
                hipEventRecord(a);
                hipLaunchKernelGGL(cal_gather_wave_local, dim3(grid), dim3(block), 0, 0, g, (uint64_t)big2, bb, iters, sink);
                hipEventRecord(b); hipEventSynchronize(b);
                float ms; hipEventElapsedTime(&ms, a, b);
                printf("{\"local\": \"wave's 64 gathers inside one random block\", \"block_bytes\": %llu, \"lane_gathers\": %.0f, \"ms\": %.4f, \"G_lane_gathers_per_s\": %.2f}\n",
                       (unsigned long long)bb, req, ms, req / ms / 1e6);
                hipEventDestroy(a); hipEventDestroy(b);
            }
        auto group = [&](int G, auto kernel) {
            hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
            hipDeviceSynchronize();
            hipEventRecord(a);
            hipLaunchKernelGGL(kernel, dim3(grid), dim3(block), 0, 0, g, (uint64_t)big2 / 32, iters, sink);
            hipEventRecord(b); hipEventSynchronize(b);
            float ms; hipEventElapsedTime(&ms, a, b);
            printf("{\"local\": \"groups of G lanes share G consecutive 32-byte records\", \"G\": %d, \"lane_gathers\": %.0f, \"ms\": %.4f, \"G_lane_gathers_per_s\": %.2f, \"G_groups_per_s\": %.2f}\n",
                   G, req, ms, req / ms / 1e6, req / G / ms / 1e6);
            hipEventDestroy(a); hipEventDestroy(b);
        };
        for (int g2 : {32, 256, 2048}) {   // workgroups of 256 lanes: 8k, 64k, 512k lanes in flight
            auto fill = [&](int mode, auto kernel) {
                hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
                hipDeviceSynchronize();
                hipEventRecord(a);
                hipLaunchKernelGGL(kernel, dim3(g2), dim3(256), 0, 0, g, (uint64_t)big2 / 128, (uint64_t)512, sink);
                hipEventRecord(b); hipEventSynchronize(b);
                float ms; hipEventElapsedTime(&ms, a, b);
                printf("{\"local\": \"line fill: random record, then 0 nothing / 1 other half of its 128-byte line / 2 another random record\", \"mode\": %d, \"lanes\": %d, \"ms\": %.4f, \"ns_per_iteration\": %.1f}\n",
                       mode, g2 * 256, ms, ms * 1e6 / 512);
                hipEventDestroy(a); hipEventDestroy(b);
            };
            fill(0, cal_line_fill<0>); fill(1, cal_line_fill<1>); fill(2, cal_line_fill<2>);
        }
        for (int rep = 0; rep < 2; ++rep) {
            group(1, cal_gather_group<1>);
            group(2, cal_gather_group<2>);
            group(4, cal_gather_group<4>);
            group(8, cal_gather_group<8>);
            group(16, cal_gather_group<16>);
        }
        return 0;
    }
    if (getenv("MICROBENCH_CAL")) {
        (void)hipFree(g);
        const size_t big2 = 3ull << 30;
        CK(hipMalloc(&g, big2)); CK(hipMemset(g, 1, big2));
        auto once = [&](const char* name, double bytes_read, double bytes_written, double requests, auto launch) {
            hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
            hipDeviceSynchronize();
            hipEventRecord(a); launch(); hipEventRecord(b); hipEventSynchronize(b);
            float ms; hipEventElapsedTime(&ms, a, b);
            printf("{\"cal\": \"%s\", \"bytes_read\": %.0f, \"bytes_written\": %.0f, \"requests\": %.0f, \"ms\": %.4f, \"G_requests_per_s\": %.2f, \"GB_per_s\": %.1f}\n",
                   name, bytes_read, bytes_written, requests, ms, requests / ms / 1e6, (bytes_read + bytes_written) / ms / 1e6);
            hipEventDestroy(a); hipEventDestroy(b);
        };
        const uint64_t iters = 128;
        const double req = (double)lanes * iters;
        hipLaunchKernelGGL(cal_stream_read, dim3(4096), dim3(256), 0, 0, g, (1ull << 30) / 16, sink);   // warm-up (first-touch of the pages)
        once("cal_stream_read", (double)(2ull << 30), 0, (double)(2ull << 30) / 64, [&] { hipLaunchKernelGGL(cal_stream_read, dim3(4096), dim3(256), 0, 0, g, (2ull << 30) / 16, sink); });
        once("cal_stream_write", 0, (double)(2ull << 30), (double)(2ull << 30) / 64, [&] { hipLaunchKernelGGL(cal_stream_write, dim3(4096), dim3(256), 0, 0, g, (2ull << 30) / 16); });
        once("cal_gather16", req * 16, 0, req, [&] { hipLaunchKernelGGL(cal_gather16, dim3(grid), dim3(block), 0, 0, g, (768ull << 20) / 16, iters, sink); });
        once("cal_gather32", req * 24, 0, req, [&] { hipLaunchKernelGGL(cal_gather32, dim3(grid), dim3(block), 0, 0, g, (1536ull << 20) / 32, iters, sink); });
        once("cal_linewrite64", 0, req * 8, req / 8, [&] { hipLaunchKernelGGL(cal_linewrite64, dim3(grid), dim3(block), 0, 0, g, big2 / 64, iters); });
        once("cal_gather_halves", req * 32, 0, req, [&] { hipLaunchKernelGGL(cal_gather_halves, dim3(grid), dim3(block), 0, 0, g, (1536ull << 20) / 128, iters, sink); });
        once("cal_gather16_plain", req * 16, 0, req, [&] { hipLaunchKernelGGL(cal_gather16_policy<0>, dim3(grid), dim3(block), 0, 0, g, (768ull << 20) / 16, iters, sink); });
        once("cal_gather16_nt", req * 16, 0, req, [&] { hipLaunchKernelGGL(cal_gather16_policy<1>, dim3(grid), dim3(block), 0, 0, g, (768ull << 20) / 16, iters, sink); });
        once("cal_gather16_sc1", req * 16, 0, req, [&] { hipLaunchKernelGGL(cal_gather16_policy<2>, dim3(grid), dim3(block), 0, 0, g, (768ull << 20) / 16, iters, sink); });
        once("cal_gather16_sc0sc1", req * 16, 0, req, [&] { hipLaunchKernelGGL(cal_gather16_policy<3>, dim3(grid), dim3(block), 0, 0, g, (768ull << 20) / 16, iters, sink); });
        once("cal_gather16_sc0sc1nt", req * 16, 0, req, [&] { hipLaunchKernelGGL(cal_gather16_policy<4>, dim3(grid), dim3(block), 0, 0, g, (768ull << 20) / 16, iters, sink); });
        {   // the same gathers from memory the L2 does not cache (MTYPE UC): does a miss then move 32 or 64 bytes instead of a line?
            uint4* uc = nullptr;
            if (hipExtMallocWithFlags((void**)&uc, 1536ull << 20, hipDeviceMallocUncached) == hipSuccess && uc) {
                hipMemset(uc, 1, 1536ull << 20);
                once("cal_gather16_uncached", req * 16, 0, req, [&] { hipLaunchKernelGGL(cal_gather16, dim3(grid), dim3(block), 0, 0, uc, (768ull << 20) / 16, iters, sink); });
                once("cal_gather32_uncached", req * 24, 0, req, [&] { hipLaunchKernelGGL(cal_gather32, dim3(grid), dim3(block), 0, 0, uc, (1536ull << 20) / 32, iters, sink); });
                (void)hipFree(uc);
            } else {
                printf("{\"cal\": \"uncached allocation failed\"}\n");
            }
            if (hipExtMallocWithFlags((void**)&uc, 1536ull << 20, hipDeviceMallocFinegrained) == hipSuccess && uc) {
                hipMemset(uc, 1, 1536ull << 20);
                once("cal_gather16_finegrained", req * 16, 0, req, [&] { hipLaunchKernelGGL(cal_gather16, dim3(grid), dim3(block), 0, 0, uc, (768ull << 20) / 16, iters, sink); });
                (void)hipFree(uc);
            }
        }
        return 0;
    }
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
    if (getenv("MICROBENCH_R2B")) {
        (void)hipFree(g);
        const size_t big2 = 3ull << 30;
        CK(hipMalloc(&g, big2)); CK(hipMemset(g, 1, big2));
        const uint64_t iters = 64;
        auto ws = [&](auto kern, int G) {
            double ms = time_ms([&] { hipLaunchKernelGGL(kern, dim3(grid), dim3(block), 0, 0, g, big2 / 16 / G, iters); });
            const double groups = (double)lanes / G * iters;
            printf("{\"bench\": \"wstore\", \"bytes_per_write\": %d, \"G_writes_per_s\": %.2f, \"G_16B_pieces_per_s\": %.2f, \"GB_per_s\": %.1f}\n", 16 * G,
                   groups / ms / 1e6, groups * G / ms / 1e6, groups * G * 16 / ms / 1e6);
        };
        ws(wstore<1>, 1); ws(wstore<2>, 2); ws(wstore<4>, 4); ws(wstore<8>, 8); ws(wstore<16>, 16);
        uint32_t* src; const uint64_t n = 46667298ull;
        CK(hipMalloc(&src, n * 4)); CK(hipMemset(src, 0, n * 4));
        double m0 = time_ms([&] { hipLaunchKernelGGL(swrite<0>, dim3(4096), dim3(256), 0, 0, g, n, src); });
        double m1 = time_ms([&] { hipLaunchKernelGGL(swrite<1>, dim3(4096), dim3(256), 0, 0, g, n, src); });
        double m2 = time_ms([&] { hipLaunchKernelGGL(swrite<2>, dim3(4096), dim3(256), 0, 0, g, n, src); });
        printf("{\"bench\": \"swrite\", \"steps\": %llu, \"ms_16B_of_32B\": %.3f, \"ms_32B\": %.3f, \"ms_16B_dense\": %.3f}\n", (unsigned long long)n, m0, m1, m2);
        return 0;
    }
    if (getenv("MICROBENCH_R2")) {
        (void)hipFree(g);
        const size_t big2 = 6ull << 30;
        CK(hipMalloc(&g, big2)); CK(hipMemset(g, 1, big2));
        for (uint64_t bytes : {768ull << 20, 1536ull << 20}) {
            const uint64_t iters = 256;
            const double total = (double)lanes * iters;
            double m16 = time_ms([&] { hipLaunchKernelGGL(gather16<1>, dim3(grid), dim3(block), 0, 0, g, bytes / 16, iters, sink); });
            double m32 = time_ms([&] { hipLaunchKernelGGL(gather32, dim3(grid), dim3(block), 0, 0, g, bytes / 32, iters, sink); });
            printf("{\"bench\": \"gather32\", \"MiB\": %llu, \"G_per_s_16B\": %.2f, \"G_per_s_32B\": %.2f}\n", (unsigned long long)(bytes >> 20), total / m16 / 1e6, total / m32 / 1e6);
        }
        for (uint64_t bytes : {1ull << 20, 16ull << 20, 128ull << 20}) {
            const uint64_t iters = 128;
            const double total = (double)lanes * iters;
            double ma = time_ms([&] { hipLaunchKernelGGL(atomics_scope<__HIP_MEMORY_SCOPE_AGENT>, dim3(grid), dim3(block), 0, 0, (unsigned long long*)g, bytes / 8, iters); });
            double mw = time_ms([&] { hipLaunchKernelGGL(atomics_scope<__HIP_MEMORY_SCOPE_WORKGROUP>, dim3(grid), dim3(block), 0, 0, (unsigned long long*)g, bytes / 8, iters); });
            printf("{\"bench\": \"atomics_scope\", \"MiB\": %llu, \"G_per_s_agent\": %.2f, \"G_per_s_workgroup\": %.2f}\n", (unsigned long long)(bytes >> 20), total / ma / 1e6, total / mw / 1e6);
        }
        unsigned int *bucket_next, *overflow;
        unsigned long long* dcoords;
        CK(hipMalloc(&bucket_next, 4096 * 4)); CK(hipMalloc(&overflow, 4));
        CK(hipMalloc(&dcoords, 1024ull * 8192 * 8)); CK(hipMemset(dcoords, 0, 1024ull * 8192 * 8));
        for (uint32_t B : {256u, 512u, 1024u}) {
            const uint64_t iters = 96;   // ~100 far messages per lane, as in one tile-kernel launch
            const double total = (double)lanes * iters;
            const uint32_t ch = 64;
            const uint32_t cap = (uint32_t)(big2 / 16 / ch / B);
            auto run = [&](auto kern, const char* name, int wg) {
                const int gr = (int)(lanes / wg);
                double best = 1e30;
                unsigned int of = 0, used = 0;
                for (int r = 0; r < 3; ++r) {
                    hipMemset(bucket_next, 0, 4096 * 4); hipMemset(overflow, 0, 4);
                    hipEvent_t a, b2; hipEventCreate(&a); hipEventCreate(&b2);
                    hipEventRecord(a);
                    hipLaunchKernelGGL(kern, dim3(gr), dim3(wg), B * 8, 0, g, bucket_next, B, cap, iters, overflow);
                    hipEventRecord(b2); hipEventSynchronize(b2);
                    float ms; hipEventElapsedTime(&ms, a, b2); if (ms < best) best = ms;
                    hipEventDestroy(a); hipEventDestroy(b2);
                }
                hipMemcpy(&of, overflow, 4, hipMemcpyDeviceToHost);
                std::vector<unsigned int> nx(B); hipMemcpy(nx.data(), bucket_next, B * 4, hipMemcpyDeviceToHost);
                for (auto v : nx) used += v;
                printf("{\"bench\": \"outbox\", \"variant\": \"%s\", \"buckets\": %u, \"wg\": %d, \"chunk_msgs\": %u, \"G_msgs_per_s\": %.2f, \"ms\": %.3f, \"overflow\": %u, \"chunks_used\": %u, \"fill\": %.3f}\n",
                       name, B, wg, ch, total / best / 1e6, best, of, used, total / ((double)used * ch));
            };
            run(outbox<64, 0>, "uniform", 256);
            run(outbox<64, 1>, "skewed", 256);
            run(outbox<64, 0>, "uniform", 1024);
            // drain of the last fill
            const uint32_t epb = 8192;
            double md = time_ms([&] { hipLaunchKernelGGL(drain<64>, dim3(B), dim3(1024), epb * 8, 0, g, bucket_next, cap, dcoords, epb); }, 2);
            printf("{\"bench\": \"drain\", \"buckets\": %u, \"ends_per_bucket\": %u, \"ms\": %.3f, \"G_msgs_per_s\": %.2f}\n", B, epb, md, total / md / 1e6);
        }
        {
            const uint64_t iters = 96;
            const double total = (double)lanes * iters;
            double ma = time_ms([&] { hipLaunchKernelGGL(outbox_atomic, dim3(grid), dim3(block), 0, 0, (unsigned long long*)g, (16ull << 20) / 8, iters); });
            printf("{\"bench\": \"outbox\", \"variant\": \"agent atomics into 16 MiB\", \"G_msgs_per_s\": %.2f}\n", total / ma / 1e6);
        }
        return 0;
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
