// host_latency.hip -- what does ONE call on host slices cost, and what would a zero-copy form cost?
// The literal trait call (one frame per call, host slice in, host slice out) is H2D -> kernel -> D2H -> sync through the
// runtime's pageable copy path.  Measured here, in C++ (no interpreter in the loop), 16 KiB frames (2048 samples):
//   a  aeth_fft_exec_host / aeth_host_vec_scale as the library runs them
//   b  an empty kernel + hipStreamSynchronize (the floor of any call that waits)
//   c  memcpy into a pinned (hipHostMalloc) bounce buffer, ONE kernel that reads and writes the pinned buffers over PCIe
//      (zero-copy), hipStreamSynchronize, memcpy out
//   d  the same with hipMemcpyAsync from / to the pinned bounce buffers around a device-resident kernel
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 -Iinclude tools/host_latency.hip -o tools/bin/host_latency \
//         -Laether_primitives_amd/lib -laether_hip -Wl,-rpath,'$ORIGIN/../../aether_primitives_amd/lib'
#include <hip/hip_runtime.h>
#include "aether_hip.h"

#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstring>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)
#define AK(x) do { int r_ = (x); if (r_) { printf("aeth error %d (%s) line %d\n", r_, aeth_last_error(), __LINE__); return 1; } } while (0)

__global__ void empty_kernel() {}
__global__ void scale_kernel(const float2 *in, float2 *out, int n, float s)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) { float2 v = in[i]; out[i] = make_float2(v.x * s, v.y * s); }
}

template <class F> static double time_us(F f, int reps = 2000)
{
    for (int i = 0; i < 50; i++) f();
    double best = 1e30;
    for (int r = 0; r < 5; r++) {
        auto t0 = std::chrono::steady_clock::now();
        for (int i = 0; i < reps; i++) f();
        best = std::min(best, std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count() / reps);
    }
    return best;
}

int main()
{
    const int n = 2048;
    aeth_ctx *ctx = nullptr; AK(aeth_ctx_create(0, &ctx));
    aeth_fft *fft = nullptr; AK(aeth_fft_create(ctx, n, 1, &fft));
    std::vector<aeth_cf32> x(n, aeth_cf32{1.f, 1.f}), y(n);
    hipStream_t s; CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    float2 *pin_in, *pin_out, *dev_in, *dev_out;
    CK(hipHostMalloc((void **)&pin_in, n * 8, hipHostMallocDefault)); CK(hipHostMalloc((void **)&pin_out, n * 8, hipHostMallocDefault));
    CK(hipMalloc((void **)&dev_in, n * 8)); CK(hipMalloc((void **)&dev_out, n * 8));

    printf("a  aeth_fft_exec_host(2048)                         %7.2f us\n", time_us([&] { aeth_fft_exec_host(fft, x.data(), n, y.data(), n, +1, 1, 0.f); }));
    printf("a  aeth_host_vec_scale(2048)                        %7.2f us\n", time_us([&] { aeth_host_vec_scale(ctx, x.data(), n, 1.0f); }));
    printf("b  empty kernel + hipStreamSynchronize              %7.2f us\n", time_us([&] { empty_kernel<<<1, 64, 0, s>>>(); (void)hipStreamSynchronize(s); }));
    printf("b' empty kernel + spin on hipStreamQuery            %7.2f us\n", time_us([&] { empty_kernel<<<1, 64, 0, s>>>(); while (hipStreamQuery(s) == hipErrorNotReady) {} }));
    printf("c  memcpy -> pinned | zero-copy kernel | sync | memcpy  %7.2f us\n", time_us([&] {
        memcpy(pin_in, x.data(), n * 8);
        scale_kernel<<<n / 256, 256, 0, s>>>(pin_in, pin_out, n, 1.0f);
        (void)hipStreamSynchronize(s);
        memcpy(y.data(), pin_out, n * 8);
    }));
    printf("c' the same, spinning on hipStreamQuery              %7.2f us\n", time_us([&] {
        memcpy(pin_in, x.data(), n * 8);
        scale_kernel<<<n / 256, 256, 0, s>>>(pin_in, pin_out, n, 1.0f);
        while (hipStreamQuery(s) == hipErrorNotReady) {}
        memcpy(y.data(), pin_out, n * 8);
    }));
    printf("d  memcpy -> pinned | H2D | kernel | D2H | sync | memcpy %7.2f us\n", time_us([&] {
        memcpy(pin_in, x.data(), n * 8);
        (void)hipMemcpyAsync(dev_in, pin_in, n * 8, hipMemcpyHostToDevice, s);
        scale_kernel<<<n / 256, 256, 0, s>>>(dev_in, dev_out, n, 1.0f);
        (void)hipMemcpyAsync(pin_out, dev_out, n * 8, hipMemcpyDeviceToHost, s);
        (void)hipStreamSynchronize(s);
        memcpy(y.data(), pin_out, n * 8);
    }));
    printf("e  pageable H2D | kernel | pageable D2H | sync (the library's shape) %7.2f us\n", time_us([&] {
        (void)hipMemcpyAsync(dev_in, x.data(), n * 8, hipMemcpyHostToDevice, s);
        scale_kernel<<<n / 256, 256, 0, s>>>(dev_in, dev_out, n, 1.0f);
        (void)hipMemcpyAsync(y.data(), dev_out, n * 8, hipMemcpyDeviceToHost, s);
        (void)hipStreamSynchronize(s);
    }));
    bool ok = true;
    for (int i = 0; i < n; i++) ok = ok && y[i].re == 1.f && y[i].im == 1.f;
    printf("zero-copy result %s\n", ok ? "ok" : "WRONG");
    return 0;
}
