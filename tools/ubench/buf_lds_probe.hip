// Probe: buffer_load_dwordx4 ... offen lds on gfx950 -- do lanes whose offset is >= num_records write ZEROS to LDS
// (hardware range check) and is soffset outside the range check?   Prints per-lane first dword landed in LDS.
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

__global__ void probe(const unsigned* __restrict__ src, const unsigned* __restrict__ voffs, unsigned num_records,
                      unsigned soff, unsigned* __restrict__ out) {
    __shared__ __attribute__((aligned(16))) unsigned lds[64 * 4];
    for (int i = threadIdx.x; i < 256; i += 64) lds[i] = 0xdeadbeefu;
    __syncthreads();
    const unsigned long long base = (unsigned long long)src;
    u32x4 srd;
    srd[0] = __builtin_amdgcn_readfirstlane((unsigned)base);
    srd[1] = __builtin_amdgcn_readfirstlane((unsigned)(base >> 32) & 0xffffu);
    srd[2] = __builtin_amdgcn_readfirstlane(num_records);
    srd[3] = 0x00020000u;
    const unsigned ldsa = __builtin_amdgcn_readfirstlane((unsigned)(size_t)(__attribute__((address_space(3))) void*)lds);
    const unsigned vo = voffs[threadIdx.x];
    const unsigned so = __builtin_amdgcn_readfirstlane(soff);
    asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tbuffer_load_dwordx4 %0, %1, %3 offen lds\n\ts_waitcnt vmcnt(0)"
                 :: "v"(vo), "s"(srd), "s"(ldsa), "s"(so) : "memory");
    __syncthreads();
    for (int i = threadIdx.x; i < 256; i += 64) out[i] = lds[i];
}

int main() {
    const int N = 4096;
    unsigned *h = new unsigned[N], *d, *dv, *dout, hv[64], ho[256];
    for (int i = 0; i < N; ++i) h[i] = 0x1000u + i;
    hipMalloc(&d, N * 4); hipMalloc(&dv, 256); hipMalloc(&dout, 1024);
    hipMemcpy(d, h, N * 4, hipMemcpyHostToDevice);
    for (int l = 0; l < 64; ++l) hv[l] = (l % 4 == 3) ? 0x80000000u : (l % 4 == 2 ? 8192u + 16u * l : 16u * l);
    hipMemcpy(dv, hv, 256, hipMemcpyHostToDevice);
    struct { unsigned nr, so; const char* name; } cases[] = {{8192u, 0u, "num_records 8192, soffset 0"}, {0x7fffffffu, 0u, "num_records 0x7fffffff"},
                                                             {8192u, 4096u, "num_records 8192, soffset 4096 (lane offsets 16 l + 4096 stay < 8192?)"}};
    for (auto& c : cases) {
        probe<<<1, 64>>>(d, dv, c.nr, c.so, dout);
        hipMemcpy(ho, dout, 1024, hipMemcpyDeviceToHost);
        printf("== %s\n", c.name);
        for (int l = 0; l < 16; ++l) printf("lane %2d voff %10u -> %08x %08x %08x %08x\n", l, hv[l], ho[4 * l], ho[4 * l + 1], ho[4 * l + 2], ho[4 * l + 3]);
    }
    return 0;
}
