// tail_probe.hip -- measurement tool (not product): how the END of a short launch train becomes visible to the host.
// bench.py's driver-style region is K = 20 step launches (~10 us each) between two synchronisation points; the
// "host tail" is what the wall clock sees on top of the HIP-event time of the train.  Variants of the closing bracket,
// each: [idle stream] t0 -> K launches of a ~10 us multi-block kernel -> <closing> -> t1, median / min over reps:
//   A  hipStreamSynchronize
//   B  one-wave signal kernel (system-scope release store to mapped pinned host memory) + host polls the word
//   C  hipStreamWriteValue64 to the same word (a command-processor packet, no kernel) + host polls
//   D  hipEventRecord + spin on hipEventQuery
//   E  as B, but the signal kernel is launched with hipExtLaunch-free plain launch on a SECOND wave of the last kernel:
//      not possible without a ticket atomics -- omitted
// Usage: tail_probe [K] [reps] [log2_elems]
#include <hip/hip_runtime.h>

#include <algorithm>
#include <chrono>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>
#if defined(__x86_64__)
#include <immintrin.h>
#endif

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } } while (0)

__global__ void __launch_bounds__(256) work_kernel(uint4 *a, const uint8_t *act, float *rew, uint8_t *term, uint32_t salt)
{
    const uint32_t i = blockIdx.x * 256 + threadIdx.x;
    uint4 v = a[i];
    uint32_t x = v.x ^ salt ^ act[i];
#pragma unroll
    for (int k = 0; k < 60; ++k)          // ~ the step's arithmetic weight
        x = x * 1664525u + 1013904223u + (x >> 7);
    v.x = x;
    a[i] = v;
    rew[i] = static_cast<float>(x & 0xffu);
    term[i] = static_cast<uint8_t>(x >> 31);
}

__global__ void __launch_bounds__(64) signal_kernel(unsigned long long *word, unsigned long long value)
{
    if (threadIdx.x == 0) {
        __threadfence_system();
        __hip_atomic_store(word, value, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    }
}

static double now_us()
{
    return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count();
}

static inline void cpu_relax()
{
#if defined(__x86_64__)
    _mm_pause();
#endif
}

int main(int argc, char **argv)
{
    const int K = argc > 1 ? atoi(argv[1]) : 20;
    const int reps = argc > 2 ? atoi(argv[2]) : 300;
    const int lg = argc > 3 ? atoi(argv[3]) : 20;
    const uint32_t n = 1u << lg;
    hipStream_t s;
    CHECK(hipStreamCreate(&s));
    uint4 *a; uint8_t *act, *term; float *rew;
    CHECK(hipMalloc(&a, n * 16ull)); CHECK(hipMalloc(&act, n)); CHECK(hipMalloc(&term, n)); CHECK(hipMalloc(&rew, n * 4ull));
    CHECK(hipMemset(a, 1, n * 16ull)); CHECK(hipMemset(act, 1, n));
    unsigned long long *word, *word_dev;
    CHECK(hipHostMalloc(reinterpret_cast<void **>(&word), 64, hipHostMallocMapped | hipHostMallocCoherent));
    CHECK(hipHostGetDevicePointer(reinterpret_cast<void **>(&word_dev), word, 0));
    *word = 0;
    hipEvent_t e0, e1, ed;
    CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1)); CHECK(hipEventCreateWithFlags(&ed, hipEventDisableTiming));
    unsigned long long seq = 0;
    const char *names[] = {"A hipStreamSynchronize", "B signal kernel + poll", "C hipStreamWriteValue64 + poll", "D event record + query spin"};
    // warm up clocks
    for (int w = 0; w < 3000; ++w)
        hipLaunchKernelGGL(work_kernel, dim3(n / 256), dim3(256), 0, s, a, act, rew, term, (uint32_t)w);
    CHECK(hipStreamSynchronize(s));
    for (int round = 0; round < 2; ++round) {
        for (int variant = 0; variant < 4; ++variant) {
            std::vector<double> wall, train;
            bool ok = true;
            for (int r = 0; r < reps && ok; ++r) {
                CHECK(hipStreamSynchronize(s));
                CHECK(hipEventRecord(e0, s));
                const double t0 = now_us();
                for (int j = 0; j < K; ++j)
                    hipLaunchKernelGGL(work_kernel, dim3(n / 256), dim3(256), 0, s, a, act, rew, term, (uint32_t)j);
                CHECK(hipEventRecord(e1, s));
                ++seq;
                switch (variant) {
                case 0: CHECK(hipStreamSynchronize(s)); break;
                case 1:
                    hipLaunchKernelGGL(signal_kernel, dim3(1), dim3(64), 0, s, word_dev, seq);
                    while (__atomic_load_n(word, __ATOMIC_ACQUIRE) < seq) cpu_relax();
                    break;
                case 2: {
                    hipError_t err = hipStreamWriteValue64(s, word_dev, seq, 0);
                    if (err != hipSuccess) { printf("%-34s not available: %s\n", names[variant], hipGetErrorString(err)); (void)hipGetLastError(); ok = false; break; }
                    while (__atomic_load_n(word, __ATOMIC_ACQUIRE) < seq) cpu_relax();
                    break;
                }
                default:
                    CHECK(hipEventRecord(ed, s));
                    while (hipEventQuery(ed) == hipErrorNotReady) cpu_relax();
                    break;
                }
                const double t1 = now_us();
                CHECK(hipStreamSynchronize(s));
                float ms = 0;
                CHECK(hipEventElapsedTime(&ms, e0, e1));
                wall.push_back(t1 - t0);
                train.push_back(ms * 1e3);
            }
            if (!ok || wall.empty())
                continue;
            std::sort(wall.begin(), wall.end());
            std::sort(train.begin(), train.end());
            const double wm = wall[wall.size() / 2], tm = train[train.size() / 2];
            printf("%-34s K=%d  wall median %7.2f us (min %7.2f)  event train median %7.2f us  tail %6.2f us  per step %6.3f us\n",
                   names[variant], K, wm, wall[0], tm, wm - tm, wm / K);
        }
    }
    return 0;
}
