#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k(unsigned *out) {
    unsigned x;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(x));
    if (threadIdx.x == 0) out[blockIdx.x] = x;
}
int main() {
    unsigned *d; hipMalloc(&d, 4096 * 4);
    k<<<4096, 64>>>(d);
    unsigned h[4096]; hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    for (int i = 0; i < 24; i++) printf("%d:%08x ", i, h[i]); printf("\n");
    int cnt[16] = {0}; int mism = 0;
    for (int i = 0; i < 4096; i++) { cnt[h[i] & 15]++; if ((h[i] & 15) != (unsigned)(i % 8)) mism++; }
    for (int i = 0; i < 16; i++) printf("xcc%d=%d ", i, cnt[i]); printf("\nmismatch vs i%%8: %d\n", mism);
}
