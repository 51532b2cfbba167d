// issue_rate.hip — how many cycles does one wave64 instruction of each class cost on a gfx950 SIMD, alone and next to
// other waves / other instruction classes?  (task 1a: settle 2 vs 4 vs 6 cycles per instruction for the
// instruction mix of the group loop of k_train_cell3.)
//
// One workgroup per CU (100 KB of LDS keeps a second one out), 256 / 512 / 1024 threads = 1 / 2 / 4 waves per SIMD.  Every
// wave runs NITER iterations of a 32-instruction unrolled body and stamps s_memtime before and after; the host prints the
// median wave's cycles per instruction and what that means per SIMD (cycles per instruction / waves per SIMD): the SIMD-level
// issue cost, the number the kernel's instruction counts have to be priced with.
//
//   hipcc --offload-arch=gfx950 -O3 -o scripts/ubench/build/issue_rate scripts/ubench/issue_rate.hip
//   scripts/ubench/build/issue_rate > profiles/r03_issue_rate.txt
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x)                                                                                   \
    do {                                                                                        \
        hipError_t e_ = (x);                                                                    \
        if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } \
    } while (0)

#define R4(x) x x x x
#define R8(x) R4(x) R4(x)
#define R32(x) R8(x) R8(x) R8(x) R8(x)

constexpr int NITER = 600;
constexpr int UNROLL = 32;
constexpr int INNER = 4;   // the body is repeated INNER times per loop iteration (loop overhead: 3 scalar instructions per 128)

enum Kind {
    K_VADD_IND, K_VADD_DEP, K_VCNDMASK, K_VCMP_CND, K_VBCNT, K_VBFE, K_DPP_SHR, K_DPP_BCAST, K_SALU, K_VALU_SALU, K_VALU2_SALU,
    K_BPERM_IND, K_BPERM_DEP, K_LDS_B32, K_LDS_B32_S6, K_LDS_B64_S6, K_LDS2_B32_S6, K_LDS_U16_RAND, K_LDS_XOR_RAND, K_LDS_XOR_RTN,
    K_READLANE, K_VALU4_LDS, K_VALU8_LDS_S2, K_SHL64, K_ADD_F64, K_FMA_F64, K_RCP_F64, K_MUL_LO, K_CVT_F64, K_POPC_CHAIN,
    K_MIX_KERNEL, K_VALU_WAITCNT, K_CND_VCC_SALU, K_CND_SGPR_SALU, K_CND_NODEP, K_CND_ADD_ALT, K_CND_1CMP3, K_CND_1CMP3_E64,
    K_CND_2STREAMS_E64, K_VADD_E64, K_VADD_LIT, K_VOP2_MIX, K_VOP3_MIX, K_VCMP_E32, K_VCMP_E64, K_VALU_SALU_3_1, K_VOP3_SALU_3_1, K_LDS_XOR_C32, K_LDS_XOR_C16, K_LDS_XOR_C4, K_LDS_XOR_SAMEBANK, K_NKIND
};
static const char *kind_name[K_NKIND] = {
    "v_add_u32 x32, 8 independent chains", "v_add_u32 x32, one dependent chain", "v_cndmask_b32 (vcc) x32",
    "v_cmp_lt_u32 + v_cndmask_b32 x16 pairs", "v_bcnt_u32_b32 x32, independent", "v_bfe_u32 x32, independent",
    "v_add_u32 dpp row_shr:1 x32, independent", "v_add_u32 dpp row_bcast:15 x32, independent", "s_add_u32 x32 (scalar only)",
    "v_add_u32 + s_add_u32 interleaved, 16 + 16", "2 v_add_u32 + 1 s_add_u32 interleaved (32 instr: 22 + 10)",
    "ds_bpermute_b32 x32 independent, one wait", "ds_bpermute_b32 x32, each waits for the last",
    "ds_read_b32 x32, lane-contiguous (conflict-free)", "ds_read_b32 x32, lane stride 6 dwords (2-way)",
    "ds_read_b64 x32, lane stride 6 dwords", "ds_read2_b32 x32, lane stride 6 dwords", "ds_read_u16 x32, random addresses",
    "ds_xor_b32 x32 (no return), random addresses", "ds_xor_rtn_b32 x32, random addresses",
    "v_readlane_b32 x32 (SGPR results)", "4 v_add_u32 + 1 ds_read_b32 (contiguous), waits every 8 reads",
    "8 v_add_u32 + 1 ds_read_b32 stride 6 + 2 s_add (group-loop-like)", "v_lshlrev_b64 x32", "v_add_f64 x32, independent",
    "v_fma_f64 x32, independent", "v_rcp_f64 x32, independent", "v_mul_lo_u32 x32, independent", "v_cvt_f64_f32 x32",
    "v_bcnt + v_cmp + v_cndmask + v_sub dependent chain (in-word select step) x8",
    "kernel mix: 24 VALU (6 cndmask, 4 cmp, 4 dpp) + 6 SALU + 3 ds_read_b32 + s_waitcnt per 34",
    "v_add_u32 x31 + s_waitcnt lgkmcnt(0) (nothing outstanding)",
    "s_mov_b64 vcc + v_cndmask_b32 (vcc) x32: mask written by the SALU", "s_mov_b64 s[20:21] + v_cndmask_b32_e64 (s[20:21]) x32",
    "v_cmp + v_cndmask_b32 (vcc) x32, destinations never read", "v_cndmask_b32 (vcc) / v_add_u32 alternating, 16 + 16",
    "(v_cmp vcc + 3 v_cndmask vcc) x8", "(v_cmp_e64 s[20:21] + 3 v_cndmask_e64 s[20:21]) x8",
    "two interleaved (v_cmp_e64 + 3 v_cndmask_e64) chains on two SGPR pairs x4", "v_add_u32_e64 (8-byte VOP3 encoding) x32",
    "v_add_u32 with a 32-bit literal (8 bytes) x32", "4-byte VOP2 mix (and, lshl, xor, sub, or, lshr, mov, min) x32",
    "8-byte VOP3 mix (lshl_add, and_or, add3, bfe, lshl_or, bcnt, mad_u24, alignbit) x32", "v_cmp_lt_u32 vcc (VOPC, 4 bytes) x32",
    "v_cmp_lt_u32_e64 s[..] (8 bytes) x32", "(3 v_add_u32 + 1 s_add_u32) x8", "(3 v_add_u32_e64 + 1 s_add_u32) x8",
    "ds_xor_b32 x32, 64 lanes on 32 distinct words (pairs share an address)", "ds_xor_b32 x32, 64 lanes on 16 distinct words",
    "ds_xor_b32 x32, 64 lanes on 4 distinct words", "ds_xor_b32 x32, 64 distinct words of ONE bank (stride 32 dwords)",
};

template <int KIND>
__global__ __launch_bounds__(1024) void k_issue(unsigned long long *cycles, unsigned *sink, int niter)
{
    extern __shared__ unsigned lds[];
    const int tid = threadIdx.x, lane = tid & 63;
    for (int i = tid; i < 16384; i += blockDim.x) lds[i] = (unsigned)i * 2654435761u;
    __syncthreads();
    unsigned a0 = tid, a1 = tid + 1, a2 = tid + 2, a3 = tid + 3, a4 = tid + 4, a5 = tid + 5, a6 = tid + 6, a7 = tid + 7;
    unsigned s0 = 1, s1 = 2;
    double d0 = 1.0 + tid, d1 = 2.0 + tid, d2 = 3.0, d3 = 4.0;
    unsigned long long q0 = tid, q1 = tid + 9;
    const unsigned wbase = (unsigned)(tid >> 6) * 4096u;                       // a 4 KB slice per wave
    unsigned addr_c = wbase + lane * 4u;                                      // contiguous dwords
    unsigned addr_s6 = wbase + lane * 24u;                                    // stride 6 dwords (8-byte aligned)
    unsigned addr_r = wbase + ((lane * 2654435761u >> 20) & 0xFFCu);           // pseudo-random dword in the slice
    unsigned addr_r16 = wbase + ((lane * 2246822519u >> 19) & 0xFFEu);         // pseudo-random u16
    unsigned bp = ((lane * 37u) & 63u) * 4u;                                   // bpermute source lane * 4
    __syncthreads();
    unsigned long long t0 = __builtin_readcyclecounter();
    __asm__ volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    for (int it = 0; it < niter; it++) {
#pragma unroll
      for (int u4 = 0; u4 < INNER; u4++) {
        if constexpr (KIND == K_VADD_IND) {
            __asm__ volatile(R4("v_add_u32 %0, %0, %8\n v_add_u32 %1, %1, %8\n v_add_u32 %2, %2, %8\n v_add_u32 %3, %3, %8\n"
                                "v_add_u32 %4, %4, %8\n v_add_u32 %5, %5, %8\n v_add_u32 %6, %6, %8\n v_add_u32 %7, %7, %8\n")
                             : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(tid));
        } else if constexpr (KIND == K_VADD_DEP) {
            __asm__ volatile(R32("v_add_u32 %0, %0, %1\n") : "+v"(a0) : "v"(tid));
        } else if constexpr (KIND == K_VCNDMASK) {
            __asm__ volatile("v_cmp_lt_u32 vcc, %8, %0\n"
                             R4("v_cndmask_b32 %0, %0, %8, vcc\n v_cndmask_b32 %1, %1, %8, vcc\n v_cndmask_b32 %2, %2, %8, vcc\n v_cndmask_b32 %3, %3, %8, vcc\n"
                                "v_cndmask_b32 %4, %4, %8, vcc\n v_cndmask_b32 %5, %5, %8, vcc\n v_cndmask_b32 %6, %6, %8, vcc\n v_cndmask_b32 %7, %7, %8, vcc\n")
                             : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(tid) : "vcc");
        } else if constexpr (KIND == K_VCMP_CND) {
            __asm__ volatile(R4("v_cmp_lt_u32 vcc, %4, %0\n v_cndmask_b32 %0, %0, %4, vcc\n v_cmp_lt_u32 vcc, %4, %1\n v_cndmask_b32 %1, %1, %4, vcc\n"
                                "v_cmp_lt_u32 vcc, %4, %2\n v_cndmask_b32 %2, %2, %4, vcc\n v_cmp_lt_u32 vcc, %4, %3\n v_cndmask_b32 %3, %3, %4, vcc\n")
                             : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(tid) : "vcc");
        } else if constexpr (KIND == K_VBCNT) {
            __asm__ volatile(R4("v_bcnt_u32_b32 %0, %8, %0\n v_bcnt_u32_b32 %1, %8, %1\n v_bcnt_u32_b32 %2, %8, %2\n v_bcnt_u32_b32 %3, %8, %3\n"
                                "v_bcnt_u32_b32 %4, %8, %4\n v_bcnt_u32_b32 %5, %8, %5\n v_bcnt_u32_b32 %6, %8, %6\n v_bcnt_u32_b32 %7, %8, %7\n")
                             : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(tid));
        } else if constexpr (KIND == K_VBFE) {
            __asm__ volatile(R4("v_bfe_u32 %0, %0, 3, 10\n v_bfe_u32 %1, %1, 3, 10\n v_bfe_u32 %2, %2, 3, 10\n v_bfe_u32 %3, %3, 3, 10\n"
                                "v_bfe_u32 %4, %4, 3, 10\n v_bfe_u32 %5, %5, 3, 10\n v_bfe_u32 %6, %6, 3, 10\n v_bfe_u32 %7, %7, 3, 10\n")
                             : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));
        } else if constexpr (KIND == K_DPP_SHR) {
            __asm__ volatile(R4("v_add_u32_dpp %0, %8, %0 row_shr:1 row_mask:0xf bank_mask:0xf\n v_add_u32_dpp %1, %8, %1 row_shr:1 row_mask:0xf bank_mask:0xf\n"
                                "v_add_u32_dpp %2, %8, %2 row_shr:1 row_mask:0xf bank_mask:0xf\n v_add_u32_dpp %3, %8, %3 row_shr:1 row_mask:0xf bank_mask:0xf\n"
                                "v_add_u32_dpp %4, %8, %4 row_shr:1 row_mask:0xf bank_mask:0xf\n v_add_u32_dpp %5, %8, %5 row_shr:1 row_mask:0xf bank_mask:0xf\n"
                                "v_add_u32_dpp %6, %8, %6 row_shr:1 row_mask:0xf bank_mask:0xf\n v_add_u32_dpp %7, %8, %7 row_shr:1 row_mask:0xf bank_mask:0xf\n")
                             : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(tid));
        } else if constexpr (KIND == K_DPP_BCAST) {
            __asm__ volatile(R4("v_add_u32_dpp %0, %8, %0 row_bcast:15 row_mask:0xa bank_mask:0xf\n v_add_u32_dpp %1, %8, %1 row_bcast:15 row_mask:0xa bank_mask:0xf\n"
                                "v_add_u32_dpp %2, %8, %2 row_bcast:15 row_mask:0xa bank_mask:0xf\n v_add_u32_dpp %3, %8, %3 row_bcast:15 row_mask:0xa bank_mask:0xf\n"
                                "v_add_u32_dpp %4, %8, %4 row_bcast:15 row_mask:0xa bank_mask:0xf\n v_add_u32_dpp %5, %8, %5 row_bcast:15 row_mask:0xa bank_mask:0xf\n"
                                "v_add_u32_dpp %6, %8, %6 row_bcast:15 row_mask:0xa bank_mask:0xf\n v_add_u32_dpp %7, %8, %7 row_bcast:15 row_mask:0xa bank_mask:0xf\n")
                             : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(tid));
        } else if constexpr (KIND == K_SALU) {
            __asm__ volatile(R8("s_add_u32 %0, %0, 3\n s_add_u32 %1, %1, 5\n s_add_u32 %0, %0, 7\n s_add_u32 %1, %1, 9\n") : "+s"(s0), "+s"(s1) :: "scc");
        } else if constexpr (KIND == K_VALU_SALU) {
            __asm__ volatile(R4("v_add_u32 %0, %0, %6\n s_add_u32 %4, %4, 3\n v_add_u32 %1, %1, %6\n s_add_u32 %5, %5, 5\n"
                                "v_add_u32 %2, %2, %6\n s_add_u32 %4, %4, 7\n v_add_u32 %3, %3, %6\n s_add_u32 %5, %5, 9\n")
                             : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+s"(s0), "+s"(s1) : "v"(tid) : "scc");
        } else if constexpr (KIND == K_VALU2_SALU) {
            __asm__ volatile(R4("v_add_u32 %0, %0, %6\n v_add_u32 %1, %1, %6\n s_add_u32 %4, %4, 3\n v_add_u32 %2, %2, %6\n"
                                "v_add_u32 %3, %3, %6\n s_add_u32 %5, %5, 5\n v_add_u32 %0, %0, %6\n v_add_u32 %1, %1, %6\n")
                             "s_add_u32 %4, %4, 3\n s_add_u32 %5, %5, 3\n"
                             : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+s"(s0), "+s"(s1) : "v"(tid) : "scc");
        } else if constexpr (KIND == K_BPERM_IND) {
            __asm__ volatile(R4("ds_bpermute_b32 %0, %8, %0\n ds_bpermute_b32 %1, %8, %1\n ds_bpermute_b32 %2, %8, %2\n ds_bpermute_b32 %3, %8, %3\n"
                                "ds_bpermute_b32 %4, %8, %4\n ds_bpermute_b32 %5, %8, %5\n ds_bpermute_b32 %6, %8, %6\n ds_bpermute_b32 %7, %8, %7\n"
                                "s_waitcnt lgkmcnt(0)\n")
                             : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(bp) : "memory");
        } else if constexpr (KIND == K_BPERM_DEP) {
            __asm__ volatile(R32("ds_bpermute_b32 %0, %1, %0\n s_waitcnt lgkmcnt(0)\n") : "+v"(a0) : "v"(bp) : "memory");
        } else if constexpr (KIND == K_LDS_B32 || KIND == K_LDS_B32_S6) {
            const unsigned ad = KIND == K_LDS_B32 ? addr_c : addr_s6;
            __asm__ volatile(R4("ds_read_b32 %0, %8\n ds_read_b32 %1, %8 offset:4\n ds_read_b32 %2, %8 offset:8\n ds_read_b32 %3, %8 offset:12\n"
                                "ds_read_b32 %4, %8 offset:16\n ds_read_b32 %5, %8 offset:20\n ds_read_b32 %6, %8 offset:256\n ds_read_b32 %7, %8 offset:260\n"
                                "s_waitcnt lgkmcnt(0)\n")
                             : "=&v"(a0), "=&v"(a1), "=&v"(a2), "=&v"(a3), "=&v"(a4), "=&v"(a5), "=&v"(a6), "=&v"(a7) : "v"(ad) : "memory");
        } else if constexpr (KIND == K_LDS_B64_S6) {
            __asm__ volatile(R4("ds_read_b64 %0, %4\n ds_read_b64 %1, %4 offset:8\n ds_read_b64 %0, %4 offset:16\n ds_read_b64 %1, %4 offset:1536\n"
                                "ds_read_b64 %2, %4 offset:1544\n ds_read_b64 %3, %4 offset:1552\n ds_read_b64 %2, %4 offset:8\n ds_read_b64 %3, %4 offset:16\n"
                                "s_waitcnt lgkmcnt(0)\n")
                             : "=&v"(q0), "=&v"(q1), "=&v"(d2), "=&v"(d3) : "v"(addr_s6) : "memory");
        } else if constexpr (KIND == K_LDS2_B32_S6) {
            __asm__ volatile(R4("ds_read2_b32 %0, %4 offset1:1\n ds_read2_b32 %1, %4 offset0:2 offset1:3\n ds_read2_b32 %0, %4 offset0:4 offset1:5\n ds_read2_b32 %1, %4 offset0:64 offset1:65\n"
                                "ds_read2_b32 %2, %4 offset0:66 offset1:67\n ds_read2_b32 %3, %4 offset0:68 offset1:69\n ds_read2_b32 %2, %4 offset0:2 offset1:3\n ds_read2_b32 %3, %4 offset0:4 offset1:5\n"
                                "s_waitcnt lgkmcnt(0)\n")
                             : "=&v"(q0), "=&v"(q1), "=&v"(d2), "=&v"(d3) : "v"(addr_s6) : "memory");
        } else if constexpr (KIND == K_LDS_U16_RAND) {
            __asm__ volatile(R4("ds_read_u16 %0, %8\n ds_read_u16 %1, %8 offset:2\n ds_read_u16 %2, %8 offset:6\n ds_read_u16 %3, %8 offset:10\n"
                                "ds_read_u16 %4, %8 offset:14\n ds_read_u16 %5, %8 offset:18\n ds_read_u16 %6, %8 offset:22\n ds_read_u16 %7, %8 offset:26\n"
                                "s_waitcnt lgkmcnt(0)\n")
                             : "=&v"(a0), "=&v"(a1), "=&v"(a2), "=&v"(a3), "=&v"(a4), "=&v"(a5), "=&v"(a6), "=&v"(a7) : "v"(addr_r16) : "memory");
        } else if constexpr (KIND == K_LDS_XOR_RAND) {
            __asm__ volatile(R4("ds_xor_b32 %0, %1\n ds_xor_b32 %0, %1 offset:4\n ds_xor_b32 %0, %1 offset:12\n ds_xor_b32 %0, %1 offset:20\n"
                                "ds_xor_b32 %0, %1 offset:28\n ds_xor_b32 %0, %1 offset:36\n ds_xor_b32 %0, %1 offset:44\n ds_xor_b32 %0, %1 offset:52\n"
                                "s_waitcnt lgkmcnt(0)\n")
                             :: "v"(addr_r), "v"(a0) : "memory");
        } else if constexpr (KIND == K_LDS_XOR_RTN) {
            __asm__ volatile(R4("ds_xor_rtn_b32 %0, %8, %9\n ds_xor_rtn_b32 %1, %8, %9 offset:4\n ds_xor_rtn_b32 %2, %8, %9 offset:12\n ds_xor_rtn_b32 %3, %8, %9 offset:20\n"
                                "ds_xor_rtn_b32 %4, %8, %9 offset:28\n ds_xor_rtn_b32 %5, %8, %9 offset:36\n ds_xor_rtn_b32 %6, %8, %9 offset:44\n ds_xor_rtn_b32 %7, %8, %9 offset:52\n"
                                "s_waitcnt lgkmcnt(0)\n")
                             : "=&v"(a0), "=&v"(a1), "=&v"(a2), "=&v"(a3), "=&v"(a4), "=&v"(a5), "=&v"(a6), "=&v"(a7) : "v"(addr_r), "v"(tid) : "memory");
        } else if constexpr (KIND == K_READLANE) {
            __asm__ volatile(R8("v_readlane_b32 %0, %2, 63\n v_readlane_b32 %1, %3, 5\n v_readlane_b32 %0, %3, 7\n v_readlane_b32 %1, %2, 9\n")
                             : "=&s"(s0), "=&s"(s1) : "v"(a0), "v"(a1));
        } else if constexpr (KIND == K_VALU4_LDS) {
            __asm__ volatile(R4("v_add_u32 %0, %0, %9\n v_add_u32 %1, %1, %9\n v_add_u32 %2, %2, %9\n v_add_u32 %3, %3, %9\n ds_read_b32 %4, %8\n"
                                "v_add_u32 %0, %0, %9\n v_add_u32 %1, %1, %9\n v_add_u32 %2, %2, %9\n v_add_u32 %3, %3, %9\n ds_read_b32 %5, %8 offset:256\n")
                             "s_waitcnt lgkmcnt(0)\n"
                             : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "=&v"(a4), "=&v"(a5), "+v"(a6), "+v"(a7) : "v"(addr_c), "v"(tid) : "memory");
        } else if constexpr (KIND == K_VALU8_LDS_S2) {
            __asm__ volatile(R4("v_add_u32 %0, %0, %9\n v_add_u32 %1, %1, %9\n v_add_u32 %2, %2, %9\n v_add_u32 %3, %3, %9\n ds_read_b32 %4, %8\n s_add_u32 %10, %10, 1\n"
                                "v_add_u32 %0, %0, %9\n v_add_u32 %1, %1, %9\n v_add_u32 %2, %2, %9\n v_add_u32 %3, %3, %9\n s_add_u32 %11, %11, 1\n")
                             "s_waitcnt lgkmcnt(0)\n"
                             : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "=&v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(addr_s6), "v"(tid), "s"(s0), "s"(s1) : "memory", "scc");
        } else if constexpr (KIND == K_SHL64) {
            __asm__ volatile(R8("v_lshlrev_b64 %0, 1, %0\n v_lshlrev_b64 %1, 1, %1\n v_lshlrev_b64 %0, 3, %0\n v_lshlrev_b64 %1, 3, %1\n") : "+v"(q0), "+v"(q1));
        } else if constexpr (KIND == K_ADD_F64) {
            __asm__ volatile(R8("v_add_f64 %0, %0, %4\n v_add_f64 %1, %1, %4\n v_add_f64 %2, %2, %4\n v_add_f64 %3, %3, %4\n")
                             : "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3) : "v"(1.0));
        } else if constexpr (KIND == K_FMA_F64) {
            __asm__ volatile(R8("v_fma_f64 %0, %0, %4, %4\n v_fma_f64 %1, %1, %4, %4\n v_fma_f64 %2, %2, %4, %4\n v_fma_f64 %3, %3, %4, %4\n")
                             : "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3) : "v"(1.0000001));
        } else if constexpr (KIND == K_RCP_F64) {
            __asm__ volatile(R8("v_rcp_f64 %0, %0\n v_rcp_f64 %1, %1\n v_rcp_f64 %2, %2\n v_rcp_f64 %3, %3\n") : "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3));
        } else if constexpr (KIND == K_MUL_LO) {
            __asm__ volatile(R8("v_mul_lo_u32 %0, %0, %4\n v_mul_lo_u32 %1, %1, %4\n v_mul_lo_u32 %2, %2, %4\n v_mul_lo_u32 %3, %3, %4\n")
                             : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(tid | 1));
        } else if constexpr (KIND == K_CVT_F64) {
            __asm__ volatile(R8("v_cvt_f64_f32 %0, %4\n v_cvt_f64_f32 %1, %5\n v_cvt_f64_f32 %2, %4\n v_cvt_f64_f32 %3, %5\n")
                             : "=&v"(d0), "=&v"(d1), "=&v"(d2), "=&v"(d3) : "v"(a0), "v"(a1));
        } else if constexpr (KIND == K_POPC_CHAIN) {
            // one bit-bisection step of the in-word select, eight times, each depending on the last
            __asm__ volatile(R8("v_and_b32 %2, 0xffff, %0\n v_bcnt_u32_b32 %2, %2, 0\n v_cmp_ge_u32 vcc, %1, %2\n v_cndmask_b32 %0, %0, %3, vcc\n")
                             : "+v"(a0), "+v"(a1), "+v"(a2) : "v"(tid) : "vcc");
        } else if constexpr (KIND == K_MIX_KERNEL) {
            __asm__ volatile(
                R4("v_add_u32 %0, %0, %9\n v_cmp_lt_u32 vcc, %9, %1\n v_cndmask_b32 %1, %1, %9, vcc\n s_add_u32 %10, %10, 1\n"
                   "v_add_u32_dpp %2, %9, %2 row_shr:1 row_mask:0xf bank_mask:0xf\n v_bcnt_u32_b32 %3, %9, %3\n"
                   "v_cndmask_b32 %5, %5, %9, vcc\n v_add_u32 %6, %6, %9\n")
                "ds_read_b32 %4, %8\n s_add_u32 %11, %11, 1\n"
                "v_cmp_lt_u32 vcc, %9, %1\n ds_read_b32 %7, %8 offset:256\n"
                : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "=&v"(a4), "+v"(a5), "+v"(a6), "=&v"(a7) : "v"(addr_s6), "v"(tid), "s"(s0), "s"(s1) : "memory", "scc", "vcc");
            __asm__ volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        } else if constexpr (KIND == K_VALU_WAITCNT) {
            __asm__ volatile(R4("v_add_u32 %0, %0, %4\n v_add_u32 %1, %1, %4\n v_add_u32 %2, %2, %4\n v_add_u32 %3, %3, %4\n"
                                "v_add_u32 %0, %0, %4\n v_add_u32 %1, %1, %4\n v_add_u32 %2, %2, %4\n s_waitcnt lgkmcnt(0)\n")
                             : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(tid));
        } else if constexpr (KIND == K_CND_VCC_SALU) {
            __asm__ volatile("s_mov_b64 vcc, 0x5555\n"
                             R4("v_cndmask_b32 %0, %0, %8, vcc\n v_cndmask_b32 %1, %1, %8, vcc\n v_cndmask_b32 %2, %2, %8, vcc\n v_cndmask_b32 %3, %3, %8, vcc\n"
                                "v_cndmask_b32 %4, %4, %8, vcc\n v_cndmask_b32 %5, %5, %8, vcc\n v_cndmask_b32 %6, %6, %8, vcc\n v_cndmask_b32 %7, %7, %8, vcc\n")
                             : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(tid) : "vcc");
        } else if constexpr (KIND == K_CND_SGPR_SALU) {
            __asm__ volatile("s_mov_b64 s[20:21], 0x5555\n"
                             R4("v_cndmask_b32_e64 %0, %0, %8, s[20:21]\n v_cndmask_b32_e64 %1, %1, %8, s[20:21]\n v_cndmask_b32_e64 %2, %2, %8, s[20:21]\n v_cndmask_b32_e64 %3, %3, %8, s[20:21]\n"
                                "v_cndmask_b32_e64 %4, %4, %8, s[20:21]\n v_cndmask_b32_e64 %5, %5, %8, s[20:21]\n v_cndmask_b32_e64 %6, %6, %8, s[20:21]\n v_cndmask_b32_e64 %7, %7, %8, s[20:21]\n")
                             : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(tid) : "s20", "s21");
        } else if constexpr (KIND == K_CND_NODEP) {
            __asm__ volatile("v_cmp_lt_u32 vcc, %8, %9\n"
                             R4("v_cndmask_b32 %0, %9, %8, vcc\n v_cndmask_b32 %1, %9, %8, vcc\n v_cndmask_b32 %2, %9, %8, vcc\n v_cndmask_b32 %3, %9, %8, vcc\n"
                                "v_cndmask_b32 %4, %9, %8, vcc\n v_cndmask_b32 %5, %9, %8, vcc\n v_cndmask_b32 %6, %9, %8, vcc\n v_cndmask_b32 %7, %9, %8, vcc\n")
                             : "=&v"(a0), "=&v"(a1), "=&v"(a2), "=&v"(a3), "=&v"(a4), "=&v"(a5), "=&v"(a6), "=&v"(a7) : "v"(tid), "v"(lane) : "vcc");
        } else if constexpr (KIND == K_CND_ADD_ALT) {
            __asm__ volatile("v_cmp_lt_u32 vcc, %8, %0\n"
                             R4("v_cndmask_b32 %0, %0, %8, vcc\n v_add_u32 %1, %1, %8\n v_cndmask_b32 %2, %2, %8, vcc\n v_add_u32 %3, %3, %8\n"
                                "v_cndmask_b32 %4, %4, %8, vcc\n v_add_u32 %5, %5, %8\n v_cndmask_b32 %6, %6, %8, vcc\n v_add_u32 %7, %7, %8\n")
                             : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(tid) : "vcc");
        } else if constexpr (KIND == K_CND_1CMP3) {
            __asm__ volatile(R8("v_cmp_lt_u32 vcc, %4, %0\n v_cndmask_b32 %0, %0, %4, vcc\n v_cndmask_b32 %1, %1, %4, vcc\n v_cndmask_b32 %2, %2, %4, vcc\n")
                             : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(tid) : "vcc");
        } else if constexpr (KIND == K_CND_1CMP3_E64) {
            __asm__ volatile(R8("v_cmp_lt_u32_e64 s[20:21], %4, %0\n v_cndmask_b32_e64 %0, %0, %4, s[20:21]\n v_cndmask_b32_e64 %1, %1, %4, s[20:21]\n v_cndmask_b32_e64 %2, %2, %4, s[20:21]\n")
                             : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(tid) : "s20", "s21");
        } else if constexpr (KIND == K_CND_2STREAMS_E64) {
            // two interleaved compare / select chains on different SGPR pairs (what a scheduler could make of the select loop)
            __asm__ volatile(R4("v_cmp_lt_u32_e64 s[20:21], %8, %0\n v_cmp_lt_u32_e64 s[22:23], %8, %4\n v_cndmask_b32_e64 %0, %0, %8, s[20:21]\n v_cndmask_b32_e64 %4, %4, %8, s[22:23]\n"
                                "v_cndmask_b32_e64 %1, %1, %8, s[20:21]\n v_cndmask_b32_e64 %5, %5, %8, s[22:23]\n v_cndmask_b32_e64 %2, %2, %8, s[20:21]\n v_cndmask_b32_e64 %6, %6, %8, s[22:23]\n")
                             : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(tid) : "s20", "s21", "s22", "s23");
        } else if constexpr (KIND == K_VADD_E64) {
            __asm__ volatile(R4("v_add_u32_e64 %0, %0, %8\n v_add_u32_e64 %1, %1, %8\n v_add_u32_e64 %2, %2, %8\n v_add_u32_e64 %3, %3, %8\n"
                                "v_add_u32_e64 %4, %4, %8\n v_add_u32_e64 %5, %5, %8\n v_add_u32_e64 %6, %6, %8\n v_add_u32_e64 %7, %7, %8\n")
                             : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(tid));
        } else if constexpr (KIND == K_VADD_LIT) {
            __asm__ volatile(R4("v_add_u32 %0, 0x12345, %0\n v_add_u32 %1, 0x12345, %1\n v_add_u32 %2, 0x12345, %2\n v_add_u32 %3, 0x12345, %3\n"
                                "v_add_u32 %4, 0x12345, %4\n v_add_u32 %5, 0x12345, %5\n v_add_u32 %6, 0x12345, %6\n v_add_u32 %7, 0x12345, %7\n")
                             : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));
        } else if constexpr (KIND == K_VOP2_MIX) {
            __asm__ volatile(R4("v_and_b32 %0, %0, %8\n v_lshlrev_b32 %1, 1, %1\n v_xor_b32 %2, %2, %8\n v_sub_u32 %3, %3, %8\n"
                                "v_or_b32 %4, %4, %8\n v_lshrrev_b32 %5, 1, %5\n v_mov_b32 %6, %8\n v_min_u32 %7, %7, %8\n")
                             : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(tid));
        } else if constexpr (KIND == K_VOP3_MIX) {
            __asm__ volatile(R4("v_lshl_add_u32 %0, %0, 1, %8\n v_and_or_b32 %1, %1, %8, %8\n v_add3_u32 %2, %2, %8, %8\n v_bfe_u32 %3, %3, 3, 10\n"
                                "v_lshl_or_b32 %4, %4, 1, %8\n v_bcnt_u32_b32 %5, %8, %5\n v_mad_u32_u24 %6, %6, %8, %8\n v_alignbit_b32 %7, %7, %8, 3\n")
                             : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(tid));
        } else if constexpr (KIND == K_VCMP_E32) {
            __asm__ volatile(R32("v_cmp_lt_u32 vcc, %0, %1\n") :: "v"(a0), "v"(a1) : "vcc");
        } else if constexpr (KIND == K_VCMP_E64) {
            __asm__ volatile(R8("v_cmp_lt_u32_e64 s[20:21], %0, %1\n v_cmp_lt_u32_e64 s[22:23], %0, %1\n v_cmp_lt_u32_e64 s[24:25], %0, %1\n v_cmp_lt_u32_e64 s[26:27], %0, %1\n")
                             :: "v"(a0), "v"(a1) : "s20", "s21", "s22", "s23", "s24", "s25", "s26", "s27");
        } else if constexpr (KIND == K_VALU_SALU_3_1) {
            __asm__ volatile(R8("v_add_u32 %0, %0, %6\n v_add_u32 %1, %1, %6\n v_add_u32 %2, %2, %6\n s_add_u32 %4, %4, 3\n")
                             : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+s"(s0), "+s"(s1) : "v"(tid) : "scc");
        } else if constexpr (KIND == K_LDS_XOR_C32 || KIND == K_LDS_XOR_C16 || KIND == K_LDS_XOR_C4 || KIND == K_LDS_XOR_SAMEBANK) {
            const unsigned nd = KIND == K_LDS_XOR_C32 ? 32u : (KIND == K_LDS_XOR_C16 ? 16u : 4u);
            const unsigned ad = KIND == K_LDS_XOR_SAMEBANK ? wbase * 0u + (unsigned)(tid >> 6) * 8u + lane * 128u   // 64 words, one bank
                                                           : wbase + (((lane * 2654435761u) >> 8) % nd) * 36u;        // nd distinct words in different banks
            __asm__ volatile(R4("ds_xor_b32 %0, %1\n ds_xor_b32 %0, %1 offset:4\n ds_xor_b32 %0, %1 offset:12\n ds_xor_b32 %0, %1 offset:20\n"
                                "ds_xor_b32 %0, %1 offset:28\n ds_xor_b32 %0, %1 offset:36\n ds_xor_b32 %0, %1 offset:44\n ds_xor_b32 %0, %1 offset:52\n"
                                "s_waitcnt lgkmcnt(0)\n")
                             :: "v"(ad), "v"(a0) : "memory");
        } else if constexpr (KIND == K_VOP3_SALU_3_1) {
            __asm__ volatile(R8("v_add_u32_e64 %0, %0, %6\n v_add_u32_e64 %1, %1, %6\n v_add_u32_e64 %2, %2, %6\n s_add_u32 %4, %4, 3\n")
                             : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+s"(s0), "+s"(s1) : "v"(tid) : "scc");
        }
      }
    }
    __asm__ volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    unsigned long long t1 = __builtin_readcyclecounter();
    if (lane == 0) cycles[blockIdx.x * (blockDim.x >> 6) + (tid >> 6)] = t1 - t0;
    // keep everything alive
    unsigned acc = a0 ^ a1 ^ a2 ^ a3 ^ a4 ^ a5 ^ a6 ^ a7 ^ s0 ^ s1 ^ (unsigned)q0 ^ (unsigned)q1 ^ (unsigned)__double2loint(d0 + d1 + d2 + d3);
    if (acc == 0x12345u) sink[0] = acc;
}

static int instr_per_body(int k)
{
    switch (k) {
    case K_CND_VCC_SALU: case K_CND_SGPR_SALU: case K_CND_NODEP: case K_CND_ADD_ALT: return 33;
    case K_VCNDMASK: return 33;
    case K_LDS_XOR_C32: case K_LDS_XOR_C16: case K_LDS_XOR_C4: case K_LDS_XOR_SAMEBANK: case K_BPERM_IND: case K_LDS_B32: case K_LDS_B32_S6: case K_LDS_B64_S6: case K_LDS2_B32_S6: case K_LDS_U16_RAND: case K_LDS_XOR_RAND: case K_LDS_XOR_RTN: return 32;  // (+ 4 waits)
    case K_BPERM_DEP: return 32;
    case K_VALU2_SALU: return 34;
    case K_VALU4_LDS: return 40;
    case K_VALU8_LDS_S2: return 44;
    case K_MIX_KERNEL: return 37;
    default: return 32;
    }
}

static int instr_per_iter(int k) { return INNER * instr_per_body(k); }

template <int KIND> static void run(unsigned long long *d_cyc, unsigned *d_sink, double clk_ghz_hint)
{
    (void)clk_ghz_hint;
    const int nblk = 256;
    CK(hipFuncSetAttribute((const void *)k_issue<KIND>, hipFuncAttributeMaxDynamicSharedMemorySize, 100 * 1024));
    printf("%-82s", kind_name[KIND]);
    for (int nt : {256, 512, 1024}) {
        const int nw = nt / 64;
        hipLaunchKernelGGL(k_issue<KIND>, dim3(nblk), dim3(nt), 100 * 1024, 0, d_cyc, d_sink, 50);  // warm-up
        hipEvent_t e0, e1;
        CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
        CK(hipEventRecord(e0, 0));
        hipLaunchKernelGGL(k_issue<KIND>, dim3(nblk), dim3(nt), 100 * 1024, 0, d_cyc, d_sink, NITER);
        CK(hipEventRecord(e1, 0));
        CK(hipDeviceSynchronize());
        float ms = 0;
        CK(hipEventElapsedTime(&ms, e0, e1));
        std::vector<unsigned long long> h((size_t)nblk * nw);
        CK(hipMemcpy(h.data(), d_cyc, h.size() * 8, hipMemcpyDeviceToHost));
        std::sort(h.begin(), h.end());
        const double med = (double)h[h.size() / 2];
        const double per_wave = med / ((double)NITER * instr_per_iter(KIND));
        const double per_simd = per_wave / (nw / 4.0);
        printf(" | %dw/SIMD: %6.2f cyc/instr/wave = %5.2f per SIMD (%.2f GHz)", nw / 4, per_wave, per_simd, med / (ms * 1e6));
        CK(hipEventDestroy(e0)); CK(hipEventDestroy(e1));
    }
    printf("\n");
    fflush(stdout);
}

template <int K> static void run_all(unsigned long long *c, unsigned *s)
{
    if constexpr (K < K_NKIND) { run<K>(c, s, 0); run_all<K + 1>(c, s); }
}

int main()
{
    unsigned long long *d_cyc;
    unsigned *d_sink;
    CK(hipMalloc(&d_cyc, 256 * 16 * 8));
    CK(hipMalloc(&d_sink, 64));
    hipDeviceProp_t p;
    CK(hipGetDeviceProperties(&p, 0));
    printf("# %s, %d CUs, issue-rate microbenchmark: %d iterations of 4 x a 32-instruction body per wave, one workgroup per CU\n", p.gcnArchName, p.multiProcessorCount, NITER);
    printf("# cyc/instr/wave: median wave's s_memtime cycles / instructions it issued; per SIMD: that / waves per SIMD (what one\n# instruction costs the SIMD when every wave runs the same stream)\n");
    run_all<0>(d_cyc, d_sink);
    return 0;
}
