// Token-pass machinery shared by the fp32 mixture kernels (forward / inverse: cnf_mixture_tok.hip, backward:
// cnf_mixture_tok_bwd.hip): pass geometry, the DMA staging of one pass of parameter spans, fixed-point helpers.
#pragma once
// cache-policy bits of the DMA loads (gfx94x/95x: 1 = sc0, 2 = nt, 16 = sc1)
#ifndef CNF_MIX_DMA_AUX
#define CNF_MIX_DMA_AUX 0
#endif
#include "cnf_mixture.h"

namespace cnf {

typedef __attribute__((address_space(3))) void lds_void_t;
typedef __attribute__((address_space(1))) const void glb_void_t;

constexpr int kMaxRowSlots = 64;       // rows one wave tile may hold
constexpr int kMaxDmaInstr = 24;       // 1 KiB DMA instructions per pass

struct TokGeom {
    int d0;             // first transformed channel
    int sd0;            // first STAGED channel: d0, or 0 when whole tokens are staged (forward / inverse only)
    int DA;             // transformed channels per token
    int lpt;            // lanes per token = DA * G
    int TPP;            // tokens per pass
    int ncopy;          // D - DA channels per token that pass through
    int contig;         // D == DA: a pass is one contiguous span
    int tokstride;      // bytes between the spans of consecutive tokens = D * P * 4
    int slot;           // LDS bytes per token slot (contig: == tokstride)
    int stage_bytes;    // LDS bytes of one wave's stage (multiple of 1 KiB)
    int tab_off;        // byte offset of the bound tables in dynamic LDS
    int acc_off;        // byte offset of the accumulator region in dynamic LDS
    int epi_off;        // byte offset of the epilogue tables + strips
    int split;          // 0: rw whole rows per wave tile; 1: S workgroups x 4 waves per row
    int rw;             // rows per wave tile (split == 0)
    int S;              // workgroups per row (split == 1)
    int ppr;            // passes per row = ceil(N / TPP)
    int pre;            // fp64 kernels: the DMA source offsets of a pass are kept in LDS (behind epi_off)
    int nt;             // forward / inverse: nontemporal DMA loads (launches that stage more than cnf_set_mixture_nt_mb megabytes)
    long ntiles;        // wave tiles (split == 0)
    FastDiv div_slot, div_n, div_lpt, div_nc;
};

// Limits of the fixed-point row sums (cnf_common.h: fix_pair_add).  A row has fewer than 65 536 terms (N * D < 65536), so terms
// below 2^14 keep its integer word below 2^30; a split row adds at most 2^14 workgroup partials, each below 2^16.
constexpr double kRowTermMax = 16384.0;
constexpr double kRowPartMax = 65536.0;

// one 1 KiB DMA instruction; nt (wave-uniform): the parameters are read once — with the nontemporal hint the fp32 forward gains 2-8 %
// from 78 MB of staged spans up (S*: 312 MB, 103.6 -> 95-100 us) and loses 4 % at configs[1]'s 52 MB (profiles/r05_mixture_nt_sweep.txt)
__device__ __forceinline__ void dma_1k(const char* gp, char* lds, int nt) {
    if (nt) __builtin_amdgcn_global_load_lds((glb_void_t*)gp, (lds_void_t*)lds, 16, 0, 2);
    else __builtin_amdgcn_global_load_lds((glb_void_t*)gp, (lds_void_t*)lds, 16, 0, CNF_MIX_DMA_AUX);
}


// Stage the parameter spans of one pass (tokens [tp, tp + npt) of the wave's tile, first span at `pass_addr`) into the
// wave's LDS stage with global_load_lds_dwordx4 (16 bytes per lane, 1 KiB per instruction; every source address is
// 16-byte aligned, chunks past the tensor's last 16 bytes are clamped onto it and never read back).  Returns the byte
// offset of lane (tli, j)'s P-float row in the stage.  The caller follows with wave_lds_sync().
__device__ __forceinline__ int stage_pass(const TokGeom& gm, char* stage_b, const char* pass_addr, const char* nn_last,
                                          int npt, int lane, int tli, int j, int P) {
    if (gm.contig) {
        const int off0 = __builtin_amdgcn_readfirstlane((int)(reinterpret_cast<uintptr_t>(pass_addr) & 15));
        const char* abase = pass_addr - off0;
        const int ni = (npt * gm.tokstride + off0 + 1023) >> 10;
        for (int i = 0; i < ni; ++i) {
            const char* gp = abase + ((size_t)(i * kWave + lane) << 4);
            gp = gp > nn_last ? nn_last : gp;
            dma_1k(gp, stage_b + (i << 10), gm.nt);
        }
        return off0 + tli * gm.tokstride + j * P * 4;
    }
    const int ni = (npt * gm.slot + 1023) >> 10;
    for (int i = 0; i < ni; ++i) {
        const uint32_t cb = (uint32_t)(i * kWave + lane) << 4;              // byte offset in the stage
        const uint32_t s = fdiv(cb, gm.div_slot);
        const uint32_t o = cb - s * (uint32_t)gm.slot;
        const char* ta = pass_addr + (size_t)s * gm.tokstride;
        const char* gp = ta - (reinterpret_cast<uintptr_t>(ta) & 15) + o;
        gp = gp > nn_last ? nn_last : gp;
        dma_1k(gp, stage_b + (i << 10), gm.nt);
    }
    const char* ta = pass_addr + (size_t)tli * gm.tokstride;
    return tli * gm.slot + (int)(reinterpret_cast<uintptr_t>(ta) & 15) + j * P * 4;
}

// host: pass geometry and work decomposition for `a`; false = shape outside what the token-pass kernels are built for
bool make_tok_geom(const MixArgs& a, int kt, int force_g, TokGeom& gm, int& G, size_t& lds, int slot_g = 0,
                   bool whole_tokens = false, long max_wgs = 0);

}  // namespace cnf
