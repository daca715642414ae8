// host_latency.hip -- measurement tool (not product): what one host-visible step of a 1-board engine can cost.
// The single-env drop-in (Game2048Env.step: one action in, ~40 bytes out) is bound by launch + completion
// latency, not by the kernel.  Variants, each timed over many iterations with the host clock:
//   A  kernel -> device buffer, hipMemcpyAsync D2H to pinned, hipStreamSynchronize      (round-2 path)
//   B  kernel writes straight into mapped pinned host memory, hipStreamSynchronize
//   C  as B, but the host polls a sequence word the kernel writes last (system-scope release), no runtime call
//   D  as C with the action passed as a kernel argument instead of being read from mapped host memory
// Usage: host_latency [iterations]
#include <hip/hip_runtime.h>

#include <chrono>
#include <cstdint>
#include <cstdio>
#include <cstdlib>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } } while (0)

struct Out {
    uint32_t cells[4];
    float reward;
    uint32_t flags;
    uint32_t pad[1];
    volatile uint32_t seq;
};

__global__ void step_like(uint4 *state, const volatile uint32_t *action_host, uint32_t action_arg, Out *out, uint32_t seq)
{
    if (threadIdx.x != 0)
        return;
    const uint32_t a = action_host ? *action_host : action_arg;
    uint4 s = *state;
    s.x = s.x * 1664525u + a + 1013904223u; // stand-in for the step arithmetic (a few hundred cycles in the real one)
    s.y ^= s.x >> 3;
    *state = s;
    out->cells[0] = s.x; out->cells[1] = s.y; out->cells[2] = s.z; out->cells[3] = s.w;
    out->reward = static_cast<float>(s.x & 0xffu);
    out->flags = a;
    if (seq) {
        __threadfence_system();
        __hip_atomic_store(const_cast<uint32_t *>(&out->seq), seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    }
}

static double now_us()
{
    return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count();
}

int main(int argc, char **argv)
{
    const int iters = argc > 1 ? atoi(argv[1]) : 20000;
    hipStream_t s;
    CHECK(hipStreamCreate(&s));
    uint4 *state;
    CHECK(hipMalloc(&state, sizeof(uint4)));
    CHECK(hipMemset(state, 0, sizeof(uint4)));
    Out *dev_out, *pinned, *pinned_dev;
    CHECK(hipMalloc(&dev_out, sizeof(Out)));
    CHECK(hipHostMalloc(reinterpret_cast<void **>(&pinned), sizeof(Out), hipHostMallocMapped | hipHostMallocCoherent));
    CHECK(hipHostGetDevicePointer(reinterpret_cast<void **>(&pinned_dev), pinned, 0));
    uint32_t *act, *act_dev;
    CHECK(hipHostMalloc(reinterpret_cast<void **>(&act), 64, hipHostMallocMapped | hipHostMallocCoherent));
    CHECK(hipHostGetDevicePointer(reinterpret_cast<void **>(&act_dev), act, 0));
    Out *host_copy;
    CHECK(hipHostMalloc(reinterpret_cast<void **>(&host_copy), sizeof(Out), 0));
    pinned->seq = 0;

    for (int variant = 0; variant < 4; ++variant) {
        uint32_t seq = 0;
        double t0 = 0;
        uint32_t sink = 0;
        for (int it = -2000; it < iters; ++it) {
            if (it == 0)
                t0 = now_us();
            *act = static_cast<uint32_t>(it) & 3u;
            switch (variant) {
            case 0:
                hipLaunchKernelGGL(step_like, dim3(1), dim3(64), 0, s, state, act_dev, 0u, dev_out, 0u);
                CHECK(hipMemcpyAsync(host_copy, dev_out, sizeof(Out), hipMemcpyDeviceToHost, s));
                CHECK(hipStreamSynchronize(s));
                sink += host_copy->cells[0];
                break;
            case 1:
                hipLaunchKernelGGL(step_like, dim3(1), dim3(64), 0, s, state, act_dev, 0u, pinned_dev, 0u);
                CHECK(hipStreamSynchronize(s));
                sink += pinned->cells[0];
                break;
            case 2:
                ++seq;
                hipLaunchKernelGGL(step_like, dim3(1), dim3(64), 0, s, state, act_dev, 0u, pinned_dev, seq);
                while (pinned->seq != seq) {
                }
                sink += pinned->cells[0];
                break;
            default:
                ++seq;
                hipLaunchKernelGGL(step_like, dim3(1), dim3(64), 0, s, state, static_cast<const uint32_t *>(nullptr), *act, pinned_dev, seq);
                while (pinned->seq != seq) {
                }
                sink += pinned->cells[0];
                break;
            }
        }
        const double us = (now_us() - t0) / iters;
        CHECK(hipStreamSynchronize(s));
        const char *names[] = {"A device out + memcpy D2H + stream sync", "B mapped pinned out + stream sync",
                               "C mapped pinned out + host polls seq", "D as C, action as kernel argument"};
        printf("%-44s %7.2f us per step  (%.0f steps/s)  [sink %u]\n", names[variant], us, 1e6 / us, sink);
    }
    return 0;
}
