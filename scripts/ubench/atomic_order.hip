// atomic_order.hip — does a returning LDS atomic (ds_add_rtn_u32) issued by ONE wave instruction serve the lanes that hit the
// same address in ascending lane order?  If it does, `old` = (counter before the instruction) + (number of lower lanes with the
// same address), and a counting sort whose counters are private to a wave is STABLE when the wave feeds it consecutive places
// in consecutive lanes — the property the sort's 7- to 9-bit passes (train_cell.hpp) rely on.  The order inside one
// instruction can only depend on the lane -> address map (the counters are private to the wave: other waves' traffic touches
// other words), so the test drives many maps through it: uniform over 1..512 bins, skewed, runs, mirrored halves, strided,
// packed 16-bit halves of one dword, with 16 waves of a block hammering their own counters at the same time.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -o scripts/ubench/build/atomic_order scripts/ubench/atomic_order.hip
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

__device__ __forceinline__ uint32_t mix(uint32_t x)
{
    x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16;
    return x;
}

// pattern p of round it for (block b, wave w, lane l): a digit in [0, nd)
__device__ __forceinline__ uint32_t digit_of(int p, uint32_t it, int b, int w, int l, uint32_t nd)
{
    const uint32_t r = mix(it * 0x9E3779B9u + (uint32_t)b * 7919u + (uint32_t)w * 104729u + (uint32_t)l * 2654435761u);
    switch (p) {
    case 0: return r % nd;                                   // uniform
    case 1: return (mix(r) % nd) * (mix(r >> 3) % nd) / nd;  // skewed towards 0
    case 2: return ((uint32_t)l / (1u + it % 7u) + it) % nd; // runs of equal digits
    case 3: return ((uint32_t)(l & 31) * (1u + it % 5u)) % nd;  // both halves of the wave alike
    case 4: return ((uint32_t)(63 - l) + r % 3u) % nd;       // descending
    case 5: return (r % 3u == 0) ? (it % nd) : (r % nd);     // a hot bin
    default: return ((uint32_t)l * (it % 64u + 1u)) % nd;    // strided
    }
}

template <bool PACKED>
__global__ __launch_bounds__(1024) void k_order(int rounds, uint32_t nd, int pattern, unsigned long long *bad, unsigned long long *total)
{
    extern __shared__ uint32_t cnt[];  // [16][row] counters, private to a wave
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const uint32_t row = (PACKED ? nd / 2 + 1 : nd) + 1;
    uint32_t *mine = cnt + w * row;
    for (uint32_t i = lane; i < row; i += 64) mine[i] = 0u;
    __syncthreads();
    unsigned long long nbad = 0;
    for (int it = 0; it < rounds; it++) {
        const uint32_t d = digit_of(pattern, (uint32_t)it, blockIdx.x, w, lane, nd);
        uint32_t before, old;
        if (PACKED) {
            before = (mine[d >> 1] >> (16 * (d & 1))) & 0xFFFFu;
            __asm__ volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            old = (atomicAdd(&mine[d >> 1], 1u << (16 * (d & 1))) >> (16 * (d & 1))) & 0xFFFFu;
        } else {
            before = mine[d];
            __asm__ volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            old = atomicAdd(&mine[d], 1u);
        }
        uint32_t below = 0;
        for (int l = 0; l < 64; l++) {
            const uint32_t dl = (uint32_t)__builtin_amdgcn_readlane((int)d, l);
            below += (l < lane && dl == d) ? 1u : 0u;
        }
        if (old != before + below) nbad++;
        if (PACKED && (it & 63) == 63) {  // keep the 16-bit halves from overflowing
            __asm__ volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            for (uint32_t i = lane; i < row; i += 64) mine[i] = 0u;
            __asm__ volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        }
    }
    if (nbad) atomicAdd(bad, nbad);
    if (tid == 0) atomicAdd(total, (unsigned long long)rounds * 1024ull);
}

int main(int argc, char **argv)
{
    const int rounds = argc > 1 ? atoi(argv[1]) : 2000, blocks = argc > 2 ? atoi(argv[2]) : 512;
    unsigned long long *d, h[2];
    CK(hipMalloc(&d, 16));
    unsigned long long all_bad = 0, all = 0;
    for (int packed = 0; packed < 2; packed++)
        for (uint32_t nd = 1; nd <= 512; nd *= 2)
            for (int p = 0; p < 7; p++) {
                if (packed && nd < 2) continue;
                CK(hipMemset(d, 0, 16));
                const size_t lds = 16 * ((packed ? nd / 2 + 1 : nd) + 1) * 4;
                if (packed) hipLaunchKernelGGL(k_order<true>, dim3(blocks), dim3(1024), lds, 0, rounds, nd, p, d, d + 1);
                else hipLaunchKernelGGL(k_order<false>, dim3(blocks), dim3(1024), lds, 0, rounds, nd, p, d, d + 1);
                CK(hipDeviceSynchronize());
                CK(hipMemcpy(h, d, 16, hipMemcpyDeviceToHost));
                if (h[0]) printf("packed=%d nd=%u pattern=%d: %llu of %llu atomics out of lane order\n", packed, nd, p, h[0], h[1]);
                all_bad += h[0]; all += h[1];
            }
    printf("ds_add_rtn lane order: %llu violations in %llu returning atomics (%s)\n", all_bad, all, all_bad ? "NOT lane-ordered" : "lane-ordered");
    return all_bad ? 2 : 0;
}
