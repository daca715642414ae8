// gap_probe.hip -- measurement tool (not product): why the first launches of a train are slow.
// A train = K back-to-back launches of a ~10 us multi-block kernel, HIP events around it.  Before each train the stream is
// synchronised and the host then waits `gap` microseconds (busy spin) before launching: 0 .. 20 000 us.  Also "no sync":
// trains follow each other with only the event records in between.  Prints the median train time per gap.
// Usage: gap_probe [K] [reps]
#include <hip/hip_runtime.h>

#include <algorithm>
#include <chrono>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } } while (0)

__global__ void __launch_bounds__(256) work_kernel(uint4 *a, const uint8_t *act, float *rew, uint8_t *term, uint32_t salt)
{
    const uint32_t i = blockIdx.x * 256 + threadIdx.x;
    uint4 v = a[i];
    uint32_t x = v.x ^ salt ^ act[i];
#pragma unroll
    for (int k = 0; k < 60; ++k)
        x = x * 1664525u + 1013904223u + (x >> 7);
    v.x = x;
    a[i] = v;
    rew[i] = static_cast<float>(x & 0xffu);
    term[i] = static_cast<uint8_t>(x >> 31);
}

static double now_us()
{
    return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count();
}

int main(int argc, char **argv)
{
    const int K = argc > 1 ? atoi(argv[1]) : 20;
    const int reps = argc > 2 ? atoi(argv[2]) : 200;
    const uint32_t n = 1u << 20;
    hipStream_t s;
    CHECK(hipStreamCreate(&s));
    uint4 *a; uint8_t *act, *term; float *rew;
    CHECK(hipMalloc(&a, n * 16ull)); CHECK(hipMalloc(&act, n)); CHECK(hipMalloc(&term, n)); CHECK(hipMalloc(&rew, n * 4ull));
    CHECK(hipMemset(a, 1, n * 16ull)); CHECK(hipMemset(act, 1, n));
    hipEvent_t e0, e1, em;
    CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1)); CHECK(hipEventCreate(&em));
    for (int w = 0; w < 3000; ++w)
        hipLaunchKernelGGL(work_kernel, dim3(n / 256), dim3(256), 0, s, a, act, rew, term, (uint32_t)w);
    CHECK(hipStreamSynchronize(s));
    const double gaps[] = {-1, 0, 20, 100, 500, 2000, 20000};
    for (double gap : gaps) {
        std::vector<double> train, first2;
        for (int r = 0; r < reps; ++r) {
            if (gap >= 0) {
                CHECK(hipStreamSynchronize(s));
                const double t = now_us();
                while (now_us() - t < gap) {
                }
            }
            CHECK(hipEventRecord(e0, s));
            for (int j = 0; j < K; ++j) {
                hipLaunchKernelGGL(work_kernel, dim3(n / 256), dim3(256), 0, s, a, act, rew, term, (uint32_t)j);
                if (j == 1)
                    CHECK(hipEventRecord(em, s));
            }
            CHECK(hipEventRecord(e1, s));
            if (gap < 0 && r + 1 < reps)
                continue; // no sync between trains: only the last one is read (events are re-recorded)
            CHECK(hipStreamSynchronize(s));
            float ms = 0, m2 = 0;
            CHECK(hipEventElapsedTime(&ms, e0, e1));
            CHECK(hipEventElapsedTime(&m2, e0, em));
            train.push_back(ms * 1e3);
            first2.push_back(m2 * 1e3);
        }
        std::sort(train.begin(), train.end());
        std::sort(first2.begin(), first2.end());
        if (gap < 0)
            printf("no sync between trains (steady state)      K=%d  train %7.2f us  per launch %6.3f   first two launches %6.2f us\n", K, train[0], train[0] / K, first2[0]);
        else
            printf("sync + %6.0f us of host idle before a train  K=%d  train median %7.2f us  per launch %6.3f   first two launches %6.2f us\n", gap, K,
                   train[train.size() / 2], train[train.size() / 2] / K, first2[first2.size() / 2]);
    }
    return 0;
}
