// snode_common.hpp -- geometry and small helpers shared by the chain-supernode translation units (snode.hip: update /
// panel / extend / pipelined substitutions; snode_g.hip: the one-pass substitution matrices and the per-level
// triangle / rows factorisation).  Included inside an anonymous namespace context of chip::dev.
#pragma once
#include "dev_common.hpp"

namespace chip {
namespace dev {
namespace {

constexpr int SN_NB = 64;
constexpr int SN_KC = 128;  // 64 * 128 * 8 = 64 KiB
constexpr int SN_U = 4;     // k-groups of A operands in flight per lane (x 2 tiles; 8 needs more than 128 registers)
constexpr int SN_WST = 4;   // k rows (entry + pivot) in flight per thread while the LDS operand is staged
constexpr int SN_WG = 512;
constexpr int SN_ROWS = 256; // panel rows per workgroup of the update kernels: 8 waves x 2 tiles of 16
typedef double snode_v4d __attribute__((ext_vector_type(4)));
typedef double snode_v2d __attribute__((ext_vector_type(2)));

// broadcast of lane `src` (a compile-time constant after unrolling) without the LDS crossbar
__device__ __forceinline__ double readlane_f64(double v, int src) {
    return __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(v), src),
                            __builtin_amdgcn_readlane(__double2loint(v), src));
}
struct SnodeGeom {
    const int *cols;
    const int *cb; // column bases of the panel (host-computed: Lp[cols[t]] - t - 1)
    double *d;     // pivots of the members, packed (k_snode_diag)
    const int8_t *sg; // signs of the members, packed
    int w, nb, h, e;
    int bn0;        // Lp[e]: where the node ids of the rows of B start in Li
    long long goff; // offset of the supernode's substitution matrix G in SnodeView::Gx, < 0: none (snode_g.hip)
};
// The supernodes of a unit level come as RECORDS in level order (`order` points at the level's first record):
// (supernode id, first member p0, width w, last member column e, rows of B, Lp[e], offset of G lo / hi) -- one 32-byte read where the
// kernels of rounds 1-2 chased order -> sn -> sn_ptr / sn_geo (three dependent loads at the head of every launch).
constexpr int SN_REC = 8;
__device__ __forceinline__ SnodeGeom snode_geom(const SnodeView &sv, const int *__restrict__ order, int idx, int &sn) {
    typedef int rec_v4i __attribute__((ext_vector_type(4)));
    const rec_v4i r0 = *(const rec_v4i *)(order + SN_REC * idx);
    const rec_v4i r1 = *(const rec_v4i *)(order + SN_REC * idx + 4);
    const int nb = r1.x;
    SnodeGeom g;
    g.bn0 = r1.y;
    g.goff = (long long)(((unsigned long long)(unsigned)r1.w << 32) | (unsigned long long)(unsigned)r1.z);
    sn = r0.x;
    g.cols = sv.sn_col + r0.y;
    g.cb = sv.sn_cb + r0.y;
    g.d = sv.sn_d + r0.y;
    g.sg = sv.sn_sg + r0.y;
    g.w = r0.z;
    g.e = r0.w;
    g.nb = nb;
    g.h = g.w + g.nb;
    return g;
}


} // namespace
} // namespace dev
} // namespace chip
