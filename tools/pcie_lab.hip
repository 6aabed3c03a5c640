// pcie_lab.hip -- what the host link gives the stream pipeline (aeth_fir_stream_host): pinned H2D alone, D2H alone,
// both at once on two streams, per chunk size.  hipcc -O2 --offload-arch=gfx950 tools/pcie_lab.hip -o tools/bin/pcie_lab
#include <hip/hip_runtime.h>

#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

int main()
{
    const size_t total = (size_t)512 << 20;
    char *hin, *hout, *din, *dout;
    CK(hipHostMalloc((void **)&hin, total, hipHostMallocDefault));
    CK(hipHostMalloc((void **)&hout, total, hipHostMallocDefault));
    CK(hipMalloc((void **)&din, total));
    CK(hipMalloc((void **)&dout, total));
    memset(hin, 1, total); memset(hout, 2, total);
    hipStream_t s0, s1;
    CK(hipStreamCreateWithFlags(&s0, hipStreamNonBlocking));
    CK(hipStreamCreateWithFlags(&s1, hipStreamNonBlocking));
    for (size_t chunk : {(size_t)4 << 20, (size_t)16 << 20, (size_t)32 << 20, (size_t)128 << 20}) {
        const size_t nch = total / chunk;
        for (int mode = 0; mode < 3; mode++) {
            double best = 1e30;
            for (int rep = 0; rep < 3; rep++) {
                CK(hipDeviceSynchronize());
                const double t0 = now();
                for (size_t k = 0; k < nch; k++) {
                    if (mode != 1) CK(hipMemcpyAsync(din + k * chunk, hin + k * chunk, chunk, hipMemcpyHostToDevice, s0));
                    if (mode != 0) CK(hipMemcpyAsync(hout + k * chunk, dout + k * chunk, chunk, hipMemcpyDeviceToHost, s1));
                }
                CK(hipStreamSynchronize(s0)); CK(hipStreamSynchronize(s1));
                const double t = now() - t0;
                if (t < best) best = t;
            }
            printf("chunk %4zu MiB  %-12s %6.1f GB/s per direction\n", chunk >> 20, mode == 0 ? "H2D" : mode == 1 ? "D2H" : "H2D + D2H", total / best / 1e9);
        }
    }
    // registering pageable memory in place: what aeth_fir_stream_host pays before its first copy
    char *pg = (char *)malloc(total);
    memset(pg, 3, total);
    double t0 = now();
    CK(hipHostRegister(pg, total, hipHostRegisterDefault));
    double t1 = now();
    CK(hipHostUnregister(pg));
    double t2 = now();
    printf("hipHostRegister of %zu MiB: %.1f ms (%.1f GB/s), unregister %.1f ms\n", total >> 20, (t1 - t0) * 1e3, total / (t1 - t0) / 1e9, (t2 - t1) * 1e3);
    // pageable copies for comparison
    t0 = now();
    CK(hipMemcpy(din, pg, total, hipMemcpyHostToDevice));
    t1 = now();
    CK(hipMemcpy(pg, dout, total, hipMemcpyDeviceToHost));
    t2 = now();
    printf("pageable hipMemcpy: H2D %.1f GB/s, D2H %.1f GB/s\n", total / (t1 - t0) / 1e9, total / (t2 - t1) / 1e9);
    return 0;
}
