// Probe of ds_read_b64_tr_b16 on gfx950: which LDS element lands in which lane / register half.
// LDS holds u16 value = element index; every lane supplies its own byte address; the four u16 results per lane are
// printed for a few address patterns.   hipcc --offload-arch=gfx950 -O3 tools/ubench/tr_b16_probe.hip -o tr_b16_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));

__global__ void probe(const unsigned* __restrict__ addr, unsigned short* __restrict__ out) {
    __shared__ __attribute__((aligned(16))) unsigned short lds[8192];
    for (int i = threadIdx.x; i < 8192; i += 64) lds[i] = (unsigned short)i;
    __syncthreads();
    const unsigned a = (unsigned)(size_t)(__attribute__((address_space(3))) void*)lds + addr[threadIdx.x];
    u32x2 r;
    asm volatile("ds_read_b64_tr_b16 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(r) : "v"(a) : "memory");
    out[threadIdx.x * 4 + 0] = (unsigned short)(r[0] & 0xffff);
    out[threadIdx.x * 4 + 1] = (unsigned short)(r[0] >> 16);
    out[threadIdx.x * 4 + 2] = (unsigned short)(r[1] & 0xffff);
    out[threadIdx.x * 4 + 3] = (unsigned short)(r[1] >> 16);
}

static void run(const char* name, unsigned (*f)(int)) {
    unsigned h[64], *d; unsigned short ho[256], *o;
    for (int l = 0; l < 64; ++l) h[l] = f(l);
    hipMalloc(&d, 256); hipMalloc(&o, 512);
    hipMemcpy(d, h, 256, hipMemcpyHostToDevice);
    probe<<<1, 64>>>(d, o);
    hipMemcpy(ho, o, 512, hipMemcpyDeviceToHost);
    printf("== %s\n", name);
    for (int l = 0; l < 64; ++l)
        printf("lane %2d addr(elem) %5u -> %5u %5u %5u %5u\n", l, h[l] / 2, ho[4 * l], ho[4 * l + 1], ho[4 * l + 2], ho[4 * l + 3]);
    hipFree(d); hipFree(o);
}

int main() {
    // A: natural [4][16] blocks per 16-lane group: lane i -> row i/4, cols 4 (i%4); group g at element 64 g
    run("A natural 4x16 per group", [](int l) -> unsigned { return 2u * ((l >> 4) * 64 + ((l & 15) >> 2) * 16 + (l & 3) * 4); });
    // B: rows 100 elements apart (pixel pitch), group g at column offset 16 g
    run("B row pitch 100, group col offset", [](int l) -> unsigned { return 2u * (((l & 15) >> 2) * 100 + (l >> 4) * 16 + (l & 3) * 4); });
    // C: every lane its own distinct block of 4: element 1000 + 8 * lane
    run("C lane-private 4 elements", [](int l) -> unsigned { return 2u * (1000 + 8 * l); });
    return 0;
}
