// l2_retain.hip -- measurement tool: does data written by one launch stay in the XCD's L2 for the next launch
// (same block -> same XCD mapping)?  A 16-byte-per-lane write kernel followed by a 16-byte-per-lane read kernel over
// the same buffer vs over another buffer, with plain and nt flavours.   Usage: l2_retain [log2_bytes=24]
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } } while (0)

typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

template <int NT> __global__ void __launch_bounds__(256) write_k(u32x4 *buf, uint32_t v)
{
    const uint32_t i = blockIdx.x * 256 + threadIdx.x;
    const u32x4 x = {v, i, v ^ i, 7u};
    if (NT) __builtin_nontemporal_store(x, buf + i); else buf[i] = x;
}

template <int NT> __global__ void __launch_bounds__(256) read_k(const u32x4 *buf, uint32_t *sink)
{
    const uint32_t i = blockIdx.x * 256 + threadIdx.x;
    const u32x4 x = NT ? __builtin_nontemporal_load(buf + i) : buf[i];
    if ((x.x ^ x.y ^ x.z ^ x.w) == 0x12345u) sink[i & 1023u] = i;   // practically never
}

int main(int argc, char **argv)
{
    const int lg = argc > 1 ? atoi(argv[1]) : 24;
    const size_t bytes = (size_t)1 << lg;
    const uint32_t n = (uint32_t)(bytes / 16);
    u32x4 *X, *Y; uint32_t *sink;
    CHECK(hipMalloc(&X, bytes)); CHECK(hipMalloc(&Y, bytes)); CHECK(hipMalloc(&sink, 4096));
    CHECK(hipMemset(Y, 1, bytes));
    hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    // pairs (write X, read X-or-Y) back to back, timed as a train: the event floor (~6 us) is paid once
    auto run = [&](const char *name, int wnt, int rnt, bool same) {
        std::vector<float> us;
        const int pairs = 100;
        for (int r = 0; r < 8; ++r) {
            CHECK(hipEventRecord(e0, 0));
            for (int j = 0; j < pairs; ++j) {
                if (wnt) hipLaunchKernelGGL(write_k<1>, dim3(n / 256), dim3(256), 0, 0, X, (uint32_t)(r + j));
                else hipLaunchKernelGGL(write_k<0>, dim3(n / 256), dim3(256), 0, 0, X, (uint32_t)(r + j));
                const u32x4 *src = same ? X : Y;
                if (rnt) hipLaunchKernelGGL(read_k<1>, dim3(n / 256), dim3(256), 0, 0, src, sink);
                else hipLaunchKernelGGL(read_k<0>, dim3(n / 256), dim3(256), 0, 0, src, sink);
            }
            CHECK(hipEventRecord(e1, 0));
            CHECK(hipEventSynchronize(e1));
            float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
            if (r >= 2) us.push_back(ms * 1e3f / pairs);
        }
        std::sort(us.begin(), us.end());
        printf("%-58s write+read pair %6.2f us (median)\n", name, us[us.size() / 2]);
    };
    printf("buffer %zu MiB\n", bytes >> 20);
    run("plain write -> plain read of the SAME buffer", 0, 0, true);
    run("plain write -> plain read of ANOTHER buffer", 0, 0, false);
    run("nt write    -> nt read of the SAME buffer", 1, 1, true);
    run("nt write    -> nt read of ANOTHER buffer", 1, 1, false);
    run("nt write    -> plain read of the SAME buffer", 1, 0, true);
    run("plain write -> nt read of the SAME buffer", 0, 1, true);
    return 0;
}
