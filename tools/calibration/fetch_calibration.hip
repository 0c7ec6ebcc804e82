// Calibration of rocprofv3's FETCH_SIZE (and the TCC_EA0_RDREQ counters behind it) for the access patterns of the BVH traversal:
// known byte counts against what the counters report.  MI355X_MICROARCH.md establishes FETCH_SIZE = 1/2 of the bytes for wide
// coalesced streaming reads; the traversal kernels gather 64 B node records (four 16 B loads per lane) and 48 B triangle records
// (three) at unrelated addresses, which the guide leaves uncalibrated.
//
//   tools/calibration/fetch_calibration [table MiB = 2048] [records per thread = 64]
// Kernels (one dispatch each per repetition, three repetitions):
//   k_cal_stream      every lane reads 16 B, lanes contiguous, the whole table once
//   k_cal_gather64    every lane reads one 64-B-aligned 64 B record per step, as 4 x raw_buffer_load_b128
//   k_cal_gather48    every lane reads one 48 B record (records packed back to back: they straddle 64 B and 128 B lines), 3 x b128
//   k_cal_gather32    32-B-aligned 32 B records, 2 x b128
//   k_cal_gather16    16 B records
// Record indices are a hash of (thread, step): uniform over the table, no locality (a 2 GiB table is 8 x Infinity Cache).
// Prints per kernel: bytes requested, time, GB/s.  Run under `rocprofv3 --kernel-trace --pmc FETCH_SIZE` and
// `--pmc TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum` (separate passes); tools/calibration/summarize.py divides.
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CHECK(x)                                                                       \
    do {                                                                               \
        hipError_t e_ = (x);                                                           \
        if (e_ != hipSuccess) {                                                        \
            std::fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_));               \
            std::exit(1);                                                              \
        }                                                                              \
    } while (0)

typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ uint32_t mix(uint32_t x) {
    x ^= x >> 16;
    x *= 0x7feb352du;
    x ^= x >> 15;
    x *= 0x846ca68bu;
    x ^= x >> 16;
    return x;
}

__global__ void __launch_bounds__(256) k_cal_stream(const uint4* table, uint64_t vec4s, uint32_t* sink) {
    uint32_t acc = 0u;
    for (uint64_t i = static_cast<uint64_t>(blockIdx.x) * blockDim.x + threadIdx.x; i < vec4s; i += static_cast<uint64_t>(gridDim.x) * blockDim.x) {
        const uint4 v = table[i];
        acc ^= v.x ^ v.y ^ v.z ^ v.w;
    }
    if (acc == 0x12345678u) sink[0] = acc;   // keeps the loads alive
}

template <int PIECES, int STRIDE>
__device__ __forceinline__ void gather(const void* table, uint32_t tableBytes, uint32_t records, uint32_t steps, uint32_t* sink) {
    const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(table), 0, tableBytes, 0x00020000);
    const uint32_t tid = blockIdx.x * blockDim.x + threadIdx.x;
    uint32_t acc = 0u, state = mix(tid * 2654435761u + 12345u);
    for (uint32_t s = 0; s < steps; ++s) {
        state = mix(state + s);
        const uint32_t record = static_cast<uint32_t>((static_cast<uint64_t>(state) * records) >> 32);
        const uint32_t at = record * STRIDE;
        u32x4 v[PIECES];
#pragma unroll
        for (int p = 0; p < PIECES; ++p) v[p] = __builtin_amdgcn_raw_buffer_load_b128(rsrc, at + 16u * p, 0, 0);   // all pieces in flight together
#pragma unroll
        for (int p = 0; p < PIECES; ++p) acc ^= v[p].x ^ v[p].y ^ v[p].z ^ v[p].w;
    }
    if (acc == 0x12345678u) sink[0] = acc;
}

__global__ void __launch_bounds__(256) k_cal_gather64(const void* t, uint32_t bytes, uint32_t records, uint32_t steps, uint32_t* sink) { gather<4, 64>(t, bytes, records, steps, sink); }
__global__ void __launch_bounds__(256) k_cal_gather48(const void* t, uint32_t bytes, uint32_t records, uint32_t steps, uint32_t* sink) { gather<3, 48>(t, bytes, records, steps, sink); }
__global__ void __launch_bounds__(256) k_cal_gather32(const void* t, uint32_t bytes, uint32_t records, uint32_t steps, uint32_t* sink) { gather<2, 32>(t, bytes, records, steps, sink); }
__global__ void __launch_bounds__(256) k_cal_gather16(const void* t, uint32_t bytes, uint32_t records, uint32_t steps, uint32_t* sink) { gather<1, 16>(t, bytes, records, steps, sink); }

int main(int argc, char** argv) {
    const uint64_t mib = argc > 1 ? std::strtoull(argv[1], nullptr, 10) : 2048ull;
    const uint32_t steps = argc > 2 ? static_cast<uint32_t>(std::atoi(argv[2])) : 64u;
    const uint64_t tableBytes = std::min<uint64_t>(mib << 20, 0xFFFFFF00ull) / 192u * 192u;   // a multiple of every record size
    void* table = nullptr;
    uint32_t* sink = nullptr;
    CHECK(hipMalloc(&table, tableBytes));
    CHECK(hipMalloc(reinterpret_cast<void**>(&sink), 64));
    CHECK(hipMemset(table, 0x5a, tableBytes));
    CHECK(hipMemset(sink, 0, 64));
    hipDeviceProp_t prop;
    CHECK(hipGetDeviceProperties(&prop, 0));
    const uint32_t blocks = static_cast<uint32_t>(prop.multiProcessorCount) * 8u, threads = blocks * 256u;
    hipEvent_t a, b;
    CHECK(hipEventCreate(&a));
    CHECK(hipEventCreate(&b));
    std::printf("table %.1f MiB, %u threads (%u blocks of 256), %u records per thread and kernel\n", tableBytes / 1048576.0, threads, blocks, steps);
    auto timed = [&](const char* name, double bytes, auto&& launch) {
        for (int rep = 0; rep < 3; ++rep) {
            CHECK(hipEventRecord(a, nullptr));
            launch();
            CHECK(hipEventRecord(b, nullptr));
            CHECK(hipEventSynchronize(b));
            CHECK(hipGetLastError());
            float ms = 0.0f;
            CHECK(hipEventElapsedTime(&ms, a, b));
            std::printf("%-16s requested %14.0f bytes  %8.3f ms  %8.1f GB/s\n", name, bytes, ms, bytes / (ms * 1e-3) / 1e9);
        }
    };
    timed("k_cal_stream", static_cast<double>(tableBytes), [&] {
        hipLaunchKernelGGL(k_cal_stream, dim3(blocks), dim3(256), 0, nullptr, static_cast<const uint4*>(table), tableBytes / 16u, sink);
    });
    const uint32_t bytes32 = static_cast<uint32_t>(tableBytes);
    timed("k_cal_gather64", 64.0 * threads * steps, [&] { hipLaunchKernelGGL(k_cal_gather64, dim3(blocks), dim3(256), 0, nullptr, table, bytes32, bytes32 / 64u, steps, sink); });
    timed("k_cal_gather48", 48.0 * threads * steps, [&] { hipLaunchKernelGGL(k_cal_gather48, dim3(blocks), dim3(256), 0, nullptr, table, bytes32, bytes32 / 48u, steps, sink); });
    timed("k_cal_gather32", 32.0 * threads * steps, [&] { hipLaunchKernelGGL(k_cal_gather32, dim3(blocks), dim3(256), 0, nullptr, table, bytes32, bytes32 / 32u, steps, sink); });
    timed("k_cal_gather16", 16.0 * threads * steps, [&] { hipLaunchKernelGGL(k_cal_gather16, dim3(blocks), dim3(256), 0, nullptr, table, bytes32, bytes32 / 16u, steps, sink); });
    CHECK(hipFree(table));
    CHECK(hipFree(sink));
    return 0;
}
