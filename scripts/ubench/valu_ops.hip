// valu_ops.hip — which gfx950 vector instructions issue at 2 cycles per wave64 instruction and which do not: every opcode the
// group loop of k_train_cell3 uses (or could use instead), 32 independent instances per body on 8 registers, at 1 / 2 / 4 waves
// per SIMD (one workgroup per CU).  Companion of issue_rate.hip; same method (s_memtime per wave, median wave).
//
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -o scripts/ubench/build/valu_ops scripts/ubench/valu_ops.hip
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
#define B8(M) M(0) M(1) M(2) M(3) M(4) M(5) M(6) M(7)
// an op on register d: %d is read and written, %8 = another VGPR, s[20:21] / vcc hold a mask
#define OP_and(d) "v_and_b32 %" #d ", %" #d ", %8\n"
#define OP_or(d) "v_or_b32 %" #d ", %" #d ", %8\n"
#define OP_xor(d) "v_xor_b32 %" #d ", %" #d ", %8\n"
#define OP_add(d) "v_add_u32 %" #d ", %" #d ", %8\n"
#define OP_sub(d) "v_sub_u32 %" #d ", %" #d ", %8\n"
#define OP_subrev(d) "v_subrev_u32 %" #d ", %" #d ", %8\n"
#define OP_mov(d) "v_mov_b32 %" #d ", %8\n"
#define OP_not(d) "v_not_b32 %" #d ", %" #d "\n"
#define OP_lshl(d) "v_lshlrev_b32 %" #d ", 3, %" #d "\n"
#define OP_lshlv(d) "v_lshlrev_b32 %" #d ", %8, %" #d "\n"
#define OP_lshr(d) "v_lshrrev_b32 %" #d ", 3, %" #d "\n"
#define OP_lshrv(d) "v_lshrrev_b32 %" #d ", %8, %" #d "\n"
#define OP_ashr(d) "v_ashrrev_i32 %" #d ", 31, %" #d "\n"
#define OP_min_u(d) "v_min_u32 %" #d ", %" #d ", %8\n"
#define OP_max_i(d) "v_max_i32 %" #d ", %" #d ", %8\n"
#define OP_min3(d) "v_min3_u32 %" #d ", %" #d ", %8, %8\n"
#define OP_med3(d) "v_med3_i32 %" #d ", %" #d ", %8, %8\n"
#define OP_add3(d) "v_add3_u32 %" #d ", %" #d ", %8, %8\n"
#define OP_lshl_add(d) "v_lshl_add_u32 %" #d ", %" #d ", 3, %8\n"
#define OP_add_lshl(d) "v_add_lshl_u32 %" #d ", %" #d ", %8, 3\n"
#define OP_lshl_or(d) "v_lshl_or_b32 %" #d ", %" #d ", 3, %8\n"
#define OP_and_or(d) "v_and_or_b32 %" #d ", %" #d ", %8, %8\n"
#define OP_or3(d) "v_or3_b32 %" #d ", %" #d ", %8, %8\n"
#define OP_xad(d) "v_xad_u32 %" #d ", %" #d ", %8, %8\n"
#define OP_bfi(d) "v_bfi_b32 %" #d ", %" #d ", %8, %8\n"
#define OP_bfe(d) "v_bfe_u32 %" #d ", %" #d ", 3, 10\n"
#define OP_bfm(d) "v_bfm_b32 %" #d ", %" #d ", %8\n"
#define OP_alignbit(d) "v_alignbit_b32 %" #d ", %" #d ", %8, 3\n"
#define OP_perm(d) "v_perm_b32 %" #d ", %" #d ", %8, %8\n"
#define OP_bcnt(d) "v_bcnt_u32_b32 %" #d ", %8, %" #d "\n"
#define OP_ffbl(d) "v_ffbl_b32 %" #d ", %" #d "\n"
#define OP_ffbh(d) "v_ffbh_u32 %" #d ", %" #d "\n"
#define OP_mbcnt(d) "v_mbcnt_lo_u32_b32 %" #d ", %8, %" #d "\n"
#define OP_mul_u24(d) "v_mul_u32_u24 %" #d ", %" #d ", %8\n"
#define OP_mad_u24(d) "v_mad_u32_u24 %" #d ", %" #d ", %8, %8\n"
#define OP_mul_lo(d) "v_mul_lo_u32 %" #d ", %" #d ", %8\n"
#define OP_mul_hi(d) "v_mul_hi_u32 %" #d ", %" #d ", %8\n"
#define OP_cnd_vcc(d) "v_cndmask_b32 %" #d ", %" #d ", %8, vcc\n"
#define OP_cnd_sgpr(d) "v_cndmask_b32_e64 %" #d ", %" #d ", %8, s[20:21]\n"
#define OP_cmp_vcc(d) "v_cmp_lt_u32 vcc, %" #d ", %8\n"
#define OP_cmp_sgpr(d) "v_cmp_lt_u32_e64 s[22:23], %" #d ", %8\n"
#define OP_cmpx(d) "v_cmpx_lt_u32 vcc, %8, %" #d "\n"
#define OP_add_co(d) "v_add_co_u32 %" #d ", vcc, %" #d ", %8\n"
#define OP_addc_co(d) "v_addc_co_u32 %" #d ", vcc, %" #d ", %8, vcc\n"
#define OP_sub_co(d) "v_sub_co_u32 %" #d ", vcc, %" #d ", %8\n"
#define OP_dpp_mov(d) "v_mov_b32_dpp %" #d ", %8 row_shr:1 row_mask:0xf bank_mask:0xf\n"
#define OP_dpp_add(d) "v_add_u32_dpp %" #d ", %8, %" #d " row_shr:1 row_mask:0xf bank_mask:0xf\n"
#define OP_dpp_ror(d) "v_add_u32_dpp %" #d ", %8, %" #d " row_ror:4 row_mask:0xf bank_mask:0xf\n"
#define OP_sdwa_add(d) "v_add_u32_sdwa %" #d ", %" #d ", %8 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_0 src1_sel:DWORD\n"
#define OP_sdwa_lshr16(d) "v_mov_b32_sdwa %" #d ", %" #d " dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1\n"
#define OP_readlane(d) "v_readlane_b32 s24, %" #d ", 5\n"
#define OP_readfirst(d) "v_readfirstlane_b32 s24, %" #d "\n"
#define OP_writelane(d) "v_writelane_b32 %" #d ", s24, 5\n"
#define OP_add_f32(d) "v_add_f32 %" #d ", %" #d ", %8\n"
#define OP_mul_f32(d) "v_mul_f32 %" #d ", %" #d ", %8\n"
#define OP_fma_f32(d) "v_fma_f32 %" #d ", %" #d ", %8, %8\n"
#define OP_fmac_f32(d) "v_fmac_f32 %" #d ", %8, %8\n"
#define OP_max_f32(d) "v_max_f32 %" #d ", %" #d ", %8\n"
#define OP_cvt_f32_u32(d) "v_cvt_f32_u32 %" #d ", %" #d "\n"
#define OP_cvt_u32_f32(d) "v_cvt_u32_f32 %" #d ", %" #d "\n"
#define OP_pk_add_u16(d) "v_pk_add_u16 %" #d ", %" #d ", %8\n"
#define OP_pk_add_f32(d) "v_pk_add_f32 %9, %9, %9\n"
#define OP_lshl_b64(d) "v_lshlrev_b64 %9, 3, %9\n"
#define OP_add_f64(d) "v_add_f64 %9, %9, %9\n"
#define OP_cmp_f32(d) "v_cmp_lt_f32 vcc, %" #d ", %8\n"
#define OP_cmp_class(d) "v_cmp_class_f32 vcc, %" #d ", %8\n"
#define OP_sad(d) "v_sad_u32 %" #d ", %" #d ", %8, %8\n"
#define OP_swap(d) "v_swap_b32 %" #d ", %8\n"
#define OP_s_add(d) "s_add_u32 s24, s24, 3\n"
#define OP_s_and64(d) "s_and_b64 s[24:25], s[24:25], s[20:21]\n"
#define OP_s_bcnt(d) "s_bcnt1_i32_b64 s24, s[20:21]\n"
#define OP_s_ff1(d) "s_ff1_i32_b64 s24, s[20:21]\n"
#define OP_s_lshr64(d) "s_lshr_b64 s[24:25], s[20:21], 3\n"
#define OP_s_nop(d) "s_nop 0\n"
// pairs: does a slow instruction next to a fast one cost the sum?
#define OP_add_bcnt(d) "v_add_u32 %" #d ", %" #d ", %8\n v_bcnt_u32_b32 %" #d ", %8, %" #d "\n"
#define OP_add_cmp(d) "v_add_u32 %" #d ", %" #d ", %8\n v_cmp_lt_u32 vcc, %" #d ", %8\n"
#define OP_add_cnd(d) "v_add_u32 %" #d ", %" #d ", %8\n v_cndmask_b32_e64 %" #d ", %" #d ", %8, s[20:21]\n"
#define OP_add_add_bcnt(d) "v_add_u32 %" #d ", %" #d ", %8\n v_and_b32 %" #d ", %" #d ", %8\n v_bcnt_u32_b32 %" #d ", %8, %" #d "\n"
#define OP_cmp_cnd(d) "v_cmp_lt_u32_e64 s[22:23], %" #d ", %8\n v_cndmask_b32_e64 %" #d ", %" #d ", %8, s[22:23]\n"
#define OP_sub_ashr_and(d) "v_sub_u32 %" #d ", %" #d ", %8\n v_ashrrev_i32 %" #d ", 31, %" #d "\n v_and_b32 %" #d ", %" #d ", %8\n"
#define OP_bcnt_salu(d) "v_bcnt_u32_b32 %" #d ", %8, %" #d "\n s_add_u32 s24, s24, 3\n"

#define OPS(X)                                                                                                                    \
    X(and, 1) X(or, 1) X(xor, 1) X(add, 1) X(sub, 1) X(subrev, 1) X(mov, 1) X(not, 1) X(lshl, 1) X(lshlv, 1) X(lshr, 1) X(lshrv, 1)  \
    X(ashr, 1) X(min_u, 1) X(max_i, 1) X(min3, 1) X(med3, 1) X(add3, 1) X(lshl_add, 1) X(add_lshl, 1) X(lshl_or, 1) X(and_or, 1)   \
    X(or3, 1) X(xad, 1) X(bfi, 1) X(bfe, 1) X(bfm, 1) X(alignbit, 1) X(perm, 1) X(bcnt, 1) X(ffbl, 1) X(ffbh, 1) X(mbcnt, 1)        \
    X(mul_u24, 1) X(mad_u24, 1) X(mul_lo, 1) X(mul_hi, 1) X(cnd_vcc, 1) X(cnd_sgpr, 1) X(cmp_vcc, 1) X(cmp_sgpr, 1) X(add_co, 1)     \
    X(addc_co, 1) X(sub_co, 1) X(dpp_mov, 1) X(dpp_add, 1) X(dpp_ror, 1) X(sdwa_add, 1) X(sdwa_lshr16, 1) X(readlane, 1)            \
    X(readfirst, 1) X(writelane, 1) X(add_f32, 1) X(mul_f32, 1) X(fma_f32, 1) X(fmac_f32, 1) X(max_f32, 1) X(cvt_f32_u32, 1)        \
    X(cvt_u32_f32, 1) X(pk_add_u16, 1) X(pk_add_f32, 1) X(lshl_b64, 1) X(add_f64, 1) X(cmp_f32, 1) X(cmp_class, 1) X(sad, 1)        \
    X(s_add, 1) X(s_and64, 1) X(s_bcnt, 1) X(s_ff1, 1) X(s_lshr64, 1) X(s_nop, 1) X(add_bcnt, 2) X(add_cmp, 2) X(add_cnd, 2)        \
    X(add_add_bcnt, 3) X(cmp_cnd, 2) X(sub_ashr_and, 3) X(bcnt_salu, 2)

enum Kind {
#define X(n, c) K_##n,
    OPS(X)
#undef X
    K_NKIND
};
static const char *kind_name[] = {
#define X(n, c) #n,
    OPS(X)
#undef X
};
static const int kind_cnt[] = {
#define X(n, c) c,
    OPS(X)
#undef X
};

constexpr int NITER = 500, INNER = 4;

template <int KIND>
__global__ __launch_bounds__(1024) void k_ops(unsigned long long *cycles, unsigned *sink, int niter)
{
    const int tid = threadIdx.x, lane = tid & 63;
    unsigned a0 = tid, a1 = tid + 1, a2 = tid + 2, a3 = tid + 3, a4 = tid + 4, a5 = tid + 5, a6 = tid + 6, a7 = tid + 7;
    unsigned b = (unsigned)tid * 2654435761u | 1u;
    double d = 1.0 + tid;
    __syncthreads();
    __asm__ volatile("s_mov_b64 s[20:21], 0x5555\n s_mov_b64 vcc, 0x3333\n s_mov_b64 s[24:25], 0" ::: "s20", "s21", "s24", "s25", "vcc");
    unsigned long long t0 = __builtin_readcyclecounter();
    __asm__ volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    for (int it = 0; it < niter; it++) {
#pragma unroll
        for (int u4 = 0; u4 < INNER; u4++) {
#define X(n, c)                                                                                                                  \
    if constexpr (KIND == K_##n)                                                                                                  \
        __asm__ volatile(R4(B8(OP_##n)) : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b), "v"(d) \
                         : "vcc", "scc", "s20", "s21", "s22", "s23", "s24", "s25");
            OPS(X)
#undef X
        }
    }
    unsigned long long t1 = __builtin_readcyclecounter();
    if (lane == 0) cycles[blockIdx.x * (blockDim.x >> 6) + (tid >> 6)] = t1 - t0;
    unsigned acc = a0 ^ a1 ^ a2 ^ a3 ^ a4 ^ a5 ^ a6 ^ a7 ^ (unsigned)__double2loint(d);
    if (acc == 0x12345u) sink[0] = acc;
}

template <int KIND> static void run(unsigned long long *d_cyc, unsigned *d_sink)
{
    const int nblk = 256;
    CK(hipFuncSetAttribute((const void *)k_ops<KIND>, hipFuncAttributeMaxDynamicSharedMemorySize, 100 * 1024));
    printf("%-14s", kind_name[KIND]);
    for (int nt : {256, 512, 1024}) {
        const int nw = nt / 64;
        hipLaunchKernelGGL(k_ops<KIND>, dim3(nblk), dim3(nt), 100 * 1024, 0, d_cyc, d_sink, 20);
        hipLaunchKernelGGL(k_ops<KIND>, dim3(nblk), dim3(nt), 100 * 1024, 0, d_cyc, d_sink, NITER);
        CK(hipDeviceSynchronize());
        std::vector<unsigned long long> h((size_t)nblk * nw);
        CK(hipMemcpy(h.data(), d_cyc, h.size() * 8, hipMemcpyDeviceToHost));
        std::sort(h.begin(), h.end());
        const double per_wave = (double)h[h.size() / 2] / ((double)NITER * INNER * 32 * kind_cnt[KIND]);
        printf("  %dw/SIMD: %6.2f per wave = %5.2f per SIMD", nw / 4, per_wave, per_wave / (nw / 4.0));
    }
    printf("\n");
    fflush(stdout);
}

template <int K> static void run_all(unsigned long long *c, unsigned *s)
{
    if constexpr (K < K_NKIND) { run<K>(c, s); run_all<K + 1>(c, s); }
}

int main()
{
    unsigned long long *d_cyc;
    unsigned *d_sink;
    CK(hipMalloc(&d_cyc, 256 * 16 * 8));
    CK(hipMalloc(&d_sink, 64));
    printf("# cycles per instruction (s_memtime, median wave; the pairs / triples at the end: per instruction of the group), one workgroup per CU\n");
    run_all<0>(d_cyc, d_sink);
    return 0;
}
