// store_hazard_repro.hip -- minimal reproduction of the fault class K1's audited two-frame instantiation showed
// (DESIGN.md section 3b).  hipcc --offload-arch=gfx950 -O2 tools/store_hazard_repro.hip -o build/bin/store_hazard_repro
//
// Sequence under test, as ROCm 7.2's hipcc emitted it in k_mog_fused<3,true,false,2> at -O2:
//       buffer_store_dwordx4 v[4:7], v_off, s[rsrc], s_soffset offen      ; 128-bit store, soffset in an SGPR
//       v_mov_b32 v5, 16                                                  ; next VALU overwrites a data register
// The GCN3..gfx9 rule "a VMEM store of more than 64 bits followed by a VALU write of its data VGPRs needs one wait
// state" carries the exemption "not when the buffer store takes its offset from an SGPR", and LLVM's hazard
// recogniser (GCNHazardRecognizer::createsVALUHazard) implements the exemption: no s_nop is inserted.  On gfx950 the
// exemption does not hold: under load the store reads its data over several cycles and the last lanes of every
// 16-lane row (12..15, 28..31, 44..47, 60..63) get the NEW register contents.
// Every lane stores {tag|4p, tag|4p+1, tag|4p+2, tag|4p+3}; a stored dword equal to 16 is a corrupted lane.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

#define NOPS0 ""
#define NOPS1 "s_nop 0\n"
#define NOPS2 "s_nop 1\n"
#define NOPS4 "s_nop 3\n"

// FORM 0: soffset in an SGPR (the exempted form); 1: soffset literal 0; 2: global_store_dwordx4 with saddr;
// 3 / 4: 64-bit and 96-bit buffer stores with an SGPR soffset (only the dwords they store are checked)
template <int FORM, int NOPS>
__global__ __launch_bounds__(256) void k_store(unsigned *out, unsigned n_lanes, unsigned soff)
{
    const unsigned p = blockIdx.x * 256u + threadIdx.x;
    const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(out, 0, (int)(n_lanes * 16u + soff), 0x00020000);
    const unsigned a = 0x80000000u | (4u * p), voff = p * 16u;
#define BODY(STORE, NOPSTR)                                                                              \
    asm volatile("v_mov_b32 v4, %0\n v_add_u32 v5, 1, %0\n v_add_u32 v6, 2, %0\n v_add_u32 v7, 3, %0\n"  \
                 STORE NOPSTR "v_mov_b32 v5, 16\n"                                                       \
                 :: "v"(a), "v"(FORM == 2 ? p * 16u : voff), "s"(rsrc), "s"(soff), "s"(out) : "v4", "v5", "v6", "v7", "memory")
#define PICK(STORE)                                                                                      \
    do { if (NOPS == 0) BODY(STORE, NOPS0); else if (NOPS == 1) BODY(STORE, NOPS1);                      \
         else if (NOPS == 2) BODY(STORE, NOPS2); else BODY(STORE, NOPS4); } while (0)
    if (FORM == 0) PICK("buffer_store_dwordx4 v[4:7], %1, %2, %3 offen\n");
    else if (FORM == 1) PICK("buffer_store_dwordx4 v[4:7], %1, %2, 0 offen\n");
    else if (FORM == 2) PICK("global_store_dwordx4 %1, v[4:7], %4\n");      // (soff == 0 in this form)
    else if (FORM == 3) PICK("buffer_store_dwordx2 v[4:5], %1, %2, %3 offen\n");  // 64-bit stores: no hazard documented
    else PICK("buffer_store_dwordx3 v[4:6], %1, %2, %3 offen\n");
}

template <int FORM, int NOPS>
static void run(const char *what, unsigned *dev, unsigned n_lanes, std::vector<unsigned> &h)
{
    long bad = 0, lanes[64] = {0};
    for (int rep = 0; rep < 5; ++rep) {
        const unsigned soff = (FORM == 0 || FORM >= 3) ? 4096u : 0u, skip = soff / 4u;
        hipMemset(dev, 0, (size_t)n_lanes * 16 + 4096);
        hipLaunchKernelGGL((k_store<FORM, NOPS>), dim3(n_lanes / 256), dim3(256), 0, 0, dev, n_lanes, soff);
        hipMemcpy(h.data(), dev + skip, (size_t)n_lanes * 16, hipMemcpyDeviceToHost);
        for (unsigned p = 0; p < n_lanes; ++p)
            for (int j = 0; j < (FORM == 3 ? 2 : FORM == 4 ? 3 : 4); ++j)
                if (h[4u * p + j] != (0x80000000u | (4u * p + j))) { ++bad; ++lanes[p & 63]; break; }
    }
    printf("%-58s wait states %d: %8ld corrupted lanes in 5 launches;", what, NOPS, bad);
    if (bad) { printf(" by lane:"); for (int l = 0; l < 64; ++l) if (lanes[l]) printf(" %d:%ld", l, lanes[l]); }
    printf("\n");
}

int main()
{
    const unsigned n_lanes = 8u << 20;         // 128 MB of 16-byte records: every CU full for many waves
    unsigned *dev = nullptr;
    if (hipMalloc((void **)&dev, (size_t)n_lanes * 16 + 4096) != hipSuccess) { fprintf(stderr, "hipMalloc failed\n"); return 2; }
    std::vector<unsigned> h((size_t)n_lanes * 4);
    run<0, 0>("buffer_store_dwordx4, soffset = SGPR (LLVM: no hazard)", dev, n_lanes, h);
    run<0, 1>("buffer_store_dwordx4, soffset = SGPR", dev, n_lanes, h);
    run<0, 2>("buffer_store_dwordx4, soffset = SGPR", dev, n_lanes, h);
    run<0, 4>("buffer_store_dwordx4, soffset = SGPR", dev, n_lanes, h);
    run<1, 0>("buffer_store_dwordx4, soffset = 0 (LLVM inserts s_nop 0)", dev, n_lanes, h);
    run<1, 1>("buffer_store_dwordx4, soffset = 0", dev, n_lanes, h);
    run<1, 2>("buffer_store_dwordx4, soffset = 0", dev, n_lanes, h);
    run<1, 4>("buffer_store_dwordx4, soffset = 0", dev, n_lanes, h);
    run<2, 0>("global_store_dwordx4 saddr (LLVM inserts s_nop 0)", dev, n_lanes, h);
    run<2, 1>("global_store_dwordx4 saddr", dev, n_lanes, h);
    run<2, 2>("global_store_dwordx4 saddr", dev, n_lanes, h);
    run<2, 4>("global_store_dwordx4 saddr", dev, n_lanes, h);
    run<3, 0>("buffer_store_dwordx2, soffset = SGPR", dev, n_lanes, h);
    run<4, 0>("buffer_store_dwordx3, soffset = SGPR", dev, n_lanes, h);
    run<4, 1>("buffer_store_dwordx3, soffset = SGPR", dev, n_lanes, h);
    hipFree(dev);
    return 0;
}
