// librt_hip.so -- gfx950 kernels + the C ABI of include/rt_abi.h.
//
// Pipeline (one "pass" = every pixel x samples_per_pass camera samples in flight):
//   k_raygen  -> [ k_trace(path rays) -> k_shade -> k_trace(shadow rays) ] x max_depth -> k_unwind -> k_accumulate
// * k_trace is a WAVE-COHERENT walk of the trace stream (the threaded pre-order BVH re-packed into filter records): the 64
//   rays of a wave advance through it together; the cursor is wave-uniform, so records arrive by scalar loads (one fetch
//   per wave).  Each lane keeps only "the next record I need" (its private skip target).  Every test is first classified
//   by an fp32 filter with a proven error bound and only undecided lanes run the reference's f64 arithmetic; subtrees that
//   provably hold no triangle / sphere the reference would ACCEPT are skipped (strict pruning).  A lane tests a subset of
//   the primitives the reference tests, in the reference's order (bvh.rs:91-106) with the reference's running t_max, and
//   never skips an accepted one: the closest hit (primitive, t) is bit-identical (DESIGN.md section 5 has the argument).
// * k_shade evaluates one ray_colour level (integrator.rs) per live path, writes a bounce record,
//   the next ray and (MIS) the shadow ray, and compacts survivors into the next queue with a
//   wave64 ballot + popcount prefix and ONE atomic per wave.
// * k_unwind applies the records last-to-first (the MIS weight gate depends on the returned
//   sub-path radiance, integrator.rs:119-136, so a running throughput cannot express it).
#include "rt_scene.h"
#include <hip/hip_runtime.h>
#include <rocprim/device/device_radix_sort.hpp>
#include <algorithm>
#include <array>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <new>
#include <string>
#include <limits>
#include <vector>
#include <thread>
#include <mutex>
#include <condition_variable>

using namespace rt;

// ------------------------------------------------------------------------------------------------
// errors
// ------------------------------------------------------------------------------------------------
static thread_local std::string g_err;
static int fail(int code, const std::string& msg) { g_err = msg; return code; }
#define HIP_OK(expr)                                                                                              \
    do {                                                                                                          \
        hipError_t e_ = (expr);                                                                                   \
        if (e_ != hipSuccess)                                                                                     \
            return fail(e_ == hipErrorOutOfMemory ? RT_ERR_OOM : RT_ERR_HIP, std::string(#expr) + ": " + hipGetErrorString(e_)); \
    } while (0)

// ------------------------------------------------------------------------------------------------
// wave helpers
// ------------------------------------------------------------------------------------------------
RT_DEV uint32_t wave_min_u32(uint32_t v) {
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) {
        uint32_t o = (uint32_t)__shfl_xor((int)v, off, 64);
        v = o < v ? o : v;
    }
    return (uint32_t)__builtin_amdgcn_readfirstlane((int)v);
}
RT_DEV uint64_t wave_sum_u64(uint64_t v) {
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) {
        uint32_t lo = (uint32_t)__shfl_xor((int)(uint32_t)v, off, 64), hi = (uint32_t)__shfl_xor((int)(uint32_t)(v >> 32), off, 64);
        v += ((uint64_t)hi << 32) | lo;
    }
    return v;
}
RT_DEV uint32_t lane_id() { return __builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u)); }

// ------------------------------------------------------------------------------------------------
// k_trace: Scene::hit (scene.rs:31-38 -> bvh.rs:91-106 / hittable.rs:41-53) for a queue of rays
// ------------------------------------------------------------------------------------------------
// The running closest hit of a lane lives in LDS, not in registers: only the exact paths read or write it, and as
// loop-carried registers these values cost the hot path a handful of copies per leaf (and four VGPRs).
template <class T> using lptr = __attribute__((address_space(3))) T*;
template <class T> struct LdsCell {
    lptr<volatile T> p;
    RT_DEV operator T() const { return *p; }
    RT_DEV LdsCell& operator=(T v) { *p = v; return *this; }
};
template <class T> struct RegCell {
    T v;
    RT_DEV operator T() const { return v; }
    RT_DEV RegCell& operator=(T x) { v = x; return *this; }
};
template <template <class> class CELL> struct LaneHitT {
    CELL<double> best;           // current t_max (bvh.rs:98-101)
    CELL<uint32_t> prim, inst;   // indices into the prims table
};
typedef LaneHitT<LdsCell> LaneHit;      // triangle scenes
typedef LaneHitT<RegCell> LaneHitRegs;  // scenes whose leaf steps are exact tests anyway (k_trace<.., REGS = true>)
struct LaneCount {
    uint32_t box, tri, sph;
    uint32_t bad;  // COUNT variant only: filter decisions contradicted by the exact test (must stay 0)
    uint32_t wsteps;  // records this wave stepped through (wave-uniform; summed into stats[15])
    uint32_t pruned_until, uncertified;  // COUNT variant only: end of the subtree strict pruning would have skipped; ray walked without certificate
#ifdef RT_TRACE_DIAG  // diagnostic build only (make diag): wave-level step counts and s_memtime cycle stamps
    uint64_t steps, leaf_steps, cyc_wait, cyc_total, lane_steps, exact_box, exact_tri, cyc_xbox, cyc_xtri, cyc_setup, cyc_leaf, leaf_open;
    uint32_t diag_flags;  // bit 0: top-level walk in the NaN-capable (literal) mode, bit 1: ray not certified for pruning, bit 2: fp32 range guard failed
#endif
};

struct GeomS { double d[9]; uint32_t kind, xform; };
RT_DEV GeomS load_geom(cptr<rt_geom> geom, uint32_t p) {
    const RT_CONST rt_geom& g = geom[p];
    GeomS r;
#pragma unroll
    for (int k = 0; k < 9; ++k) r.d[k] = g.d[k];
    r.kind = g.kind; r.xform = g.xform;
    return r;
}
// Gives the join of per-lane (divergent) branches a block of its own.  Without it the compiler folds that join into the
// next wave-uniform join (ultimately the loop header), and every scalar merged there -- the cursor, the cached
// transform id -- is classed divergent and moves to VGPRs with exec-masked branches.
#define RT_RECONVERGE() asm volatile("")
#define RT_RAY_STRIDE 8u  /* a ray in the path buffers is one 64-byte line: origin, direction, time, pad (two double4) */
struct __attribute__((aligned(16))) PathHit { double t; uint32_t prim, inst; };            // closest hit of a path's current ray
struct __attribute__((aligned(16))) PathInfo { uint32_t pixel, sample, nb, pad; };         // nb: bounce records written

// ---- the trace stream ---------------------------------------------------------------------------
// rt_scene_upload derives from the blob's node / geom tables one pre-order byte stream of FILTER records:
//     header, 32 B:  fp32 box as (min, max) pairs per axis | skip_off | flags(4) + record bytes/16 (4) + node index (24)
//     + for a leaf of one triangle (block repeats it) or two under one transform (RT_STREAM_TRI2), 112 B:
//       xform id | pad | 12 fp32 (A, B) pairs: e2.xyz, ml.xyz = v0 x e2, -n.xyz = -(e1 x e2), E2', ML', N'' (error weights)
//     + for a leaf of one or two spheres under one transform (RT_STREAM_SPH2), 80 B: xform id | count | (centre, radius) f64 x 2
//     + for any other leaf (RT_STREAM_LEAF), 16 B: prim0 | prim1 | xform of prim0 | pad
// Every range (the scene-level array, each instance sub-array) is closed by an END record -- a list-item header that
// every ray passes -- so the walk needs no bounds test: skip links to the end of a range land on it.
// A traversal step loads one window of the stream (two s_load_dwordx16 + one dwordx2 issued back to back,
// ONE wait): the header, the triangle block if any, and the line after it, which holds the records the
// walk visits next when it continues sequentially.  The loads are inline asm because the compiler sinks
// ordinary loads to their first use, which re-serialises them.
//
// The stream only feeds the fp32 FILTERS below.  A filter answers "certainly the reference's decision is
// pass / fail / reject" or "unknown"; unknown lanes are decided by the exact f64 code on the blob's own
// tables (nodes[node index], geom[prim]), which is the reference's arithmetic operation for operation.
// So results never depend on the filters -- only on their error bounds being valid, which the COUNT
// variant of the kernel re-checks against the exact test for every decision (LaneCount::bad).
/* bits 2-3 of the header flags (stream only): what the record holds */
#define RT_STREAM_KIND 12u
#define RT_STREAM_INNER 0u
#define RT_STREAM_TRI2 4u      /* leaf of one triangle, or two under the same transform, with fp32 filter block */
#define RT_STREAM_LEAF 8u      /* any other leaf: exact tests on the blob's own tables */
#define RT_STREAM_SPH2 12u     /* leaf of one or two spheres under the same transform, f64 geometry inline */
#define RT_STREAM_REC_SPH2 112u
#define RT_STREAM_REC_LEAF 48u     /* header + prim0 | prim1 | xform of prim0 | pad: the exact path starts its geometry loads without a detour through rt_node */
#define RT_STREAM_REC 32u
#define RT_STREAM_REC_INNER 64u    /* header + local-space box of the subtree (strict pruning, see StrictFilter) + pad */
#define RT_STREAM_REC_TRI2 144u
#define RT_STREAM_PAD 384u     /* readable bytes past the last record: the window over-reads */
#define RT_STREAM_END_NODE 0xFFFFFFu  /* node field of the END record that closes every range (scene array, each instance sub-array) */
#define RT_STREAM_MAX_NODES RT_STREAM_END_NODE
typedef uint32_t u32x16 __attribute__((ext_vector_type(16)));
typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
typedef float f2 __attribute__((ext_vector_type(2)));
struct Window { u32x16 a, b; u32x2 c; };
RT_DEV void window_issue(Window& w, uint64_t base, uint32_t off) {  // address = base + off + imm: no per-step address arithmetic
    asm volatile("s_load_dwordx16 %0, %3, %4 offset:0x0\n\t"
                 "s_load_dwordx16 %1, %3, %4 offset:0x40\n\t"
                 "s_load_dwordx2 %2, %3, %4 offset:0x80"
                 : "=&s"(w.a), "=&s"(w.b), "=&s"(w.c) : "s"(base), "s"(off));
}
// The wait directly follows the issue: a destination register of a load still in flight must never be left to the
// register allocator (a spill or re-use before the data lands would be silently overwritten) -- which is also why
// there is no fire-and-forget prefetch of the skip target here (tried: it measured slower as well).
RT_DEV void window_wait(Window& w) { asm volatile("s_waitcnt lgkmcnt(0)" : "+s"(w.a), "+s"(w.b), "+s"(w.c)); }
template <int K> RT_DEV uint32_t wword(const Window& w) {
    if constexpr (K < 16) return w.a[K];
    else if constexpr (K < 32) return w.b[K - 16];
    else return w.c[K - 32];
}
RT_DEV double sdbl(uint32_t lo, uint32_t hi) { return __builtin_bit_cast(double, ((uint64_t)hi << 32) | (uint64_t)lo); }
template <int K> RT_DEV uint64_t wpair(const Window& w) { return ((uint64_t)wword<K + 1>(w) << 32) | (uint64_t)wword<K>(w); }  // dwords K, K+1 as one SGPR pair
template <int J> RT_DEV uint64_t tri_pair(const Window& w) { return wpair<10 + 2 * J>(w); }

// ---- packed fp32: one instruction evaluates the (A, B) or (min, max) pair held in an SGPR pair against a per-lane
// value taken from the low (SEL 0) or high (SEL 1) half of a VGPR pair.  op_sel picks the source half for the low
// result, op_sel_hi for the high result.
#define PK_MUL(r, s, v, SEL) asm("v_pk_mul_f32 %0, %1, %2 op_sel:[0," #SEL "] op_sel_hi:[1," #SEL "]" : "=v"(r) : "s"(s), "v"(v))
#define PK_FMA_ACC(r, s, v, SEL, acc) asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[0," #SEL ",0] op_sel_hi:[1," #SEL ",1]" : "=v"(r) : "s"(s), "v"(v), "v"(acc))
#define PK_FMA_BB(r, s, v, SEL, c, CSEL) asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[0," #SEL "," #CSEL "] op_sel_hi:[1," #SEL "," #CSEL "]" : "=v"(r) : "s"(s), "v"(v), "v"(c))
RT_DEV float vmin_f32(float a, float b) { float r; asm("v_min_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b)); return r; }
RT_DEV float vmax_f32(float a, float b) { float r; asm("v_max_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b)); return r; }
RT_DEV float vmax3_f32(float a, float b, float c) { float r; asm("v_max3_f32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c)); return r; }
RT_DEV float vmin3_f32(float a, float b, float c) { float r; asm("v_min3_f32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c)); return r; }

// Lane predicates of the hot path are kept as 64-bit wave masks in SGPR pairs: the compares write them there directly,
// the scalar unit combines them, and "any lane?" is a scalar compare.  (A ballot of a C++ bool costs a
// v_cndmask + v_cmp each time.)  Valid only where every lane of the wave is enabled, which holds throughout the walk.
typedef uint64_t lanemask;
#define RT_CMP_F32(NAME, OP) RT_DEV lanemask NAME(float a, float b) { lanemask m; asm("v_cmp_" OP "_f32_e64 %0, %1, %2" : "=s"(m) : "v"(a), "v"(b)); return m; }
RT_CMP_F32(mask_le, "le") RT_CMP_F32(mask_ge, "ge") RT_CMP_F32(mask_gt, "gt") RT_CMP_F32(mask_lt, "lt")
RT_DEV lanemask mask_ge_const(float a, float s) { lanemask m; asm("v_cmp_ge_f32_e64 %0, %1, %2" : "=s"(m) : "v"(a), "s"(s)); return m; }
RT_DEV lanemask mask_neg(float a) { lanemask m; asm("v_cmp_gt_f32_e64 %0, 0, %1" : "=s"(m) : "v"(a)); return m; }
RT_DEV lanemask mask_reached(uint32_t off, uint32_t wait) { lanemask m; asm("v_cmp_ge_u32_e64 %0, %1, %2" : "=s"(m) : "s"(off), "v"(wait)); return m; }
RT_DEV lanemask mask_uniform(lanemask m) {  // a no-op for a mask the compiler already holds in SGPRs; the variants with the nested walk sometimes class it divergent
    return ((uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(m >> 32)) << 32) | (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)m);
}
RT_DEV uint32_t mask_select(uint32_t a, uint32_t b, lanemask m) { uint32_t r; m = mask_uniform(m); asm("v_cndmask_b32_e64 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "s"(m)); return r; }  // m ? b : a
RT_DEV bool mask_lane(lanemask m) { return mask_select(0u, 1u, m) != 0u; }  // this lane's bit (exact paths only)

// double -> float with the result strictly on the stated side (RN conversion errs by <= 2^-24 |x|)
RT_DEV float f32_below(double v) {
    if (!(fabs(v) <= 3.0e38)) return v > 0.0 ? 3.0e38f : -__builtin_inff();  // NaN: -inf ("nothing is certainly below")
    const float f = (float)v;
    return f - fabsf(f) * 0x1p-22f - 1e-30f;
}
RT_DEV float f32_above(double v) {
    if (!(fabs(v) <= 3.0e38)) return v < 0.0 ? -3.0e38f : __builtin_inff();
    const float f = (float)v;
    return f + fabsf(f) * 0x1p-22f + 1e-30f;
}

// ---- box filter --------------------------------------------------------------------------------------------------
// Exact decision (box_axis_fast, PLAIN rays): pass <=> max_k near_k <= best  and  min_k far_k >= t_min, with
// near_k / far_k = min / max of the reference's f64 values t = RN(RN(b - o) * inv) for the two planes of axis k.
// Filter: t' = fma32(b', inv', c') with b' = RN32(b), inv' = RN32(inv), c' = RN32(RN64(-o * inv)).  For finite b:
//   |t' - t| <= 2^-24 (2.0001 |b inv| + 1.0001 |o inv| + |t|)(1 + 2^-23)  <  0.76 * 2^-22 (|b| + |o|) |inv|
// so with B >= |b| for every finite box coordinate of the scene,  M = max_k 2^-22 (B + |o_k|) |inv_k|  (rounded up, plus
// an absolute 1e-30 for fp32 underflow) bounds the error of every plane value, hence of nearmax = max_k min(t'_lo, t'_hi)
// and farmin = min_k max(t'_lo, t'_hi) (max / min are 1-Lipschitz).  Therefore
//   nearmax' <= best - M  and  farmin' >= t_min + M   =>  the exact test passes      (thresholds rounded inward)
//   nearmax' >  best + M  or   farmin' <  t_min - M   =>  the exact test fails       (thresholds rounded outward)
// and anything else (including every NaN) is "unknown".  fp32 subnormal inputs would add errors the bound does not
// cover, so rays with an |inv_k| or |o_k| outside [1e-30, 1e30] or a bound above 1e30 (and scenes with B outside that range)
// get M = +inf: never certain.
// List items carry a (-inf, +inf) box: t' = -inf / +inf exactly as in f64, "certain pass" (PLAIN rays cannot fail them).
struct BoxFilter {
    f2 inv_xy, invz_cx, c_yz;  // (inv.x, inv.y) (inv.z, c.x) (c.y, c.z)
    float M, hiP, hiF, loP, loF;
};
RT_DEV void box_filter_set_best(BoxFilter& f, double best) { f.hiP = f32_below(best - (double)f.M); f.hiF = f32_above(best + (double)f.M); }
RT_DEV void box_filter_init(BoxFilter& f, d3 o, d3 inv, double t_min, double best, double box_abs_max) {
    const double cx = -(o.x * inv.x), cy = -(o.y * inv.y), cz = -(o.z * inv.z);
    f.inv_xy = (f2){(float)inv.x, (float)inv.y}; f.invz_cx = (f2){(float)inv.z, (float)cx}; f.c_yz = (f2){(float)cy, (float)cz};
    const double ax = fabs(inv.x), ay = fabs(inv.y), az = fabs(inv.z), ox = fabs(o.x), oy = fabs(o.y), oz = fabs(o.z);
    const bool in_range = fmin(fmin(ax, ay), az) >= 1e-30 && fmax(fmax(ax, ay), az) <= 1e30 && fmax(fmax(ox, oy), oz) <= 1e30 &&
                          box_abs_max >= 1e-30 && box_abs_max <= 1e30;
    const double m = fmax(fmax((box_abs_max + ox) * ax, (box_abs_max + oy) * ay), (box_abs_max + oz) * az) * 0x1p-22;
    // m <= 1e30 keeps every b' inv' and c' far below the fp32 overflow threshold: no inf - inf, hence no NaN for a min / max to drop
    f.M = (in_range && m <= 1e30) ? f32_above(m) : __builtin_inff();
    box_filter_set_best(f, best);
    f.loP = f32_above(t_min + (double)f.M); f.loF = f32_below(t_min - (double)f.M);
}
// returns certain-pass / certain-fail lane masks
template <int B>  // B: first dword of the record's header inside the window
RT_DEV void box_filter(const BoxFilter& f, const Window& w, lanemask& cp, lanemask& cf) {
    f2 tx, ty, tz;
    PK_FMA_BB(tx, wpair<B + 0>(w), f.inv_xy, 0, f.invz_cx, 1);
    PK_FMA_BB(ty, wpair<B + 2>(w), f.inv_xy, 1, f.c_yz, 0);
    PK_FMA_BB(tz, wpair<B + 4>(w), f.invz_cx, 0, f.c_yz, 1);
    // one asm block: separate asm statements each cost a conservative hazard s_nop, and the scalar unit is the busy one
    float n0, n1, n2, f0, f1, f2;
    lanemask m0, m1, m2, m3;
    asm("v_min_f32 %[n0], %[ax], %[bx]\n\tv_max_f32 %[f0], %[ax], %[bx]\n\t"
        "v_min_f32 %[n1], %[ay], %[by]\n\tv_max_f32 %[f1], %[ay], %[by]\n\t"
        "v_min_f32 %[n2], %[az], %[bz]\n\tv_max_f32 %[f2], %[az], %[bz]\n\t"
        "v_max3_f32 %[n0], %[n0], %[n1], %[n2]\n\t"   // nearmax
        "v_min3_f32 %[f0], %[f0], %[f1], %[f2]\n\t"   // farmin
        "v_cmp_le_f32_e64 %[m0], %[n0], %[hiP]\n\tv_cmp_ge_f32_e64 %[m1], %[f0], %[loP]\n\t"
        "v_cmp_gt_f32_e64 %[m2], %[n0], %[hiF]\n\tv_cmp_lt_f32_e64 %[m3], %[f0], %[loF]"
        : [n0] "=&v"(n0), [n1] "=&v"(n1), [n2] "=&v"(n2), [f0] "=&v"(f0), [f1] "=&v"(f1), [f2] "=&v"(f2),
          [m0] "=&s"(m0), [m1] "=&s"(m1), [m2] "=&s"(m2), [m3] "=&s"(m3)
        : [ax] "v"(tx.x), [bx] "v"(tx.y), [ay] "v"(ty.x), [by] "v"(ty.y), [az] "v"(tz.x), [bz] "v"(tz.y),
          [hiP] "v"(f.hiP), [loP] "v"(f.loP), [hiF] "v"(f.hiF), [loF] "v"(f.loF));
    cp = m0 & m1;
    cf = m2 | m3;
}

// ---- triangle filter -----------------------------------------------------------------------------------------------
// The reference rejects a triangle unless u = RN(RN(1/det) * sh) lies in [0, 1], where det = e1.(d x e2) and
// sh = (o - v0).(d x e2) are its rounded f64 values (mesh.rs:57-66).  In real arithmetic det = d.(-n) with n = e1 x e2 and
// sh = e2.(o x d) + d.(v0 x e2).  The filter evaluates these two dot products in fp32 for both triangles of the leaf
// at once (mo = RN32(RN64(o x d)), ml = RN32(RN64(v0 x e2)), n likewise from the host) and bounds the distance to the
// reference's values:
//   |sh' - sh_ref|  <= 8 * 2^-24 (|e2|_inf |mo|_1 + |d|_inf |ml|_1)  +  f64 roundings <= 2^-48 (|o|_inf + |v0|_inf) |d|_inf |e2|_1
//   |det' - det_ref| <= 5 * 2^-24 |d|_inf |n|_1                        +  f64 roundings <= 2^-48 |d|_inf |e1|_1 |e2|_1
// (6- and 3-term fp32 dot products by FMA: gamma_6 + 2u per product, Hoelder for the sums).  The stream carries
//   E2' = 2^-20 |e2|_inf,  ML' = 2^-20 |ml|_1 + 2^-46 |v0|_inf |e2|_1,  N'' = (1 + 2^-19)(2^-20 |n|_1 + 2^-46 |e1|_1 |e2|_1)
// (each rounded up) and the ray  D = |d|_inf,  MO' = |mo|_1 + 2^-26 |o|_inf |d|_inf  (rounded up), so that
//   ms = E2' MO' + D ML' >= 2 x the sh bound,   md = D N'' >= 2 x (1 + 2^-19) x the det bound.
// With a = |sh'| - ms <= |sh_ref|, c = |det'| - md <= |det_ref|, b = RN32(|det'| (1 + 2^-20) + md) >= (1 + 2^-40)|det_ref|:
//   (R1') a >= 2e-18, c >= 2e-18, sh' det' < 0  =>  sh_ref and det_ref have the signs of sh', det' (opposite), and
//         sh_ref * det_ref <= -4e-36: premise R1 of tri_stage_a (rt_scene.h) holds unless |det_ref| < EPS  => reject;
//   (R2') a >= 2e-18, a > b                     =>  |sh_ref| > (1 + 2^-40)|det_ref|: premise R2 holds unless |det_ref| < EPS => reject
// (|det_ref| <= |det'| + md is finite in fp32, so the 1e100 guard of tri_stage_a holds; |det_ref| < EPS is the
// reference's own rejection).  The 2e-18 floors keep fp32 underflow (absolute errors <= 1e-37) out of both tests;
// fp32 overflow gives inf - inf = NaN and no rejection.  Everything not rejected goes to the exact f64 test.
struct TriFilter { f2 d_xy, dz_mox, mo_yz, D_MO; };
RT_DEV void tri_filter_init(TriFilter& f, d3 o, d3 d) {
    const d3 mo = cross(o, d);
    f.d_xy = (f2){(float)d.x, (float)d.y}; f.dz_mox = (f2){(float)d.z, (float)mo.x}; f.mo_yz = (f2){(float)mo.y, (float)mo.z};
    const double D = fmax(fmax(fabs(d.x), fabs(d.y)), fabs(d.z)), O = fmax(fmax(fabs(o.x), fabs(o.y)), fabs(o.z));
    f.D_MO = (f2){f32_above(D), f32_above((fabs(mo.x) + fabs(mo.y)) + fabs(mo.z) + 0x1p-26 * O * D)};
}
RT_DEV lanemask tri_filter_one(float sh, float det, float ms, float md) {  // set: certainly rejected by the reference
    const float a = fabsf(sh) - ms, c = fabsf(det) - md;
    const float b = __builtin_fmaf(fabsf(det), 1.0f + 0x1p-20f, md);
    const float p = sh * det;
    return mask_ge_const(a, 2e-18f) & (mask_gt(a, b) | (mask_ge_const(c, 2e-18f) & mask_neg(p)));
}
RT_DEV void tri_filter(const TriFilter& f, const Window& w, lanemask& rejA, lanemask& rejB) {
    f2 sh, det, ms, md;
    // one asm block (see box_filter); op_sel picks the low / high half of the per-lane pair for both triangles
    asm("v_pk_mul_f32 %[sh], %[e2x], %[dzmx] op_sel:[0,1] op_sel_hi:[1,1]\n\t"            // e2.x * mo.x
        "v_pk_mul_f32 %[det], %[nx], %[dxy] op_sel:[0,0] op_sel_hi:[1,0]\n\t"             // -n.x * d.x
        "v_pk_mul_f32 %[ms], %[ML], %[DMO] op_sel:[0,0] op_sel_hi:[1,0]\n\t"              // ML' * D
        "v_pk_mul_f32 %[md], %[NN], %[DMO] op_sel:[0,0] op_sel_hi:[1,0]\n\t"              // N'' * D
        "v_pk_fma_f32 %[sh], %[e2y], %[moyz], %[sh] op_sel:[0,0,0] op_sel_hi:[1,0,1]\n\t"  // e2.y * mo.y
        "v_pk_fma_f32 %[det], %[ny], %[dxy], %[det] op_sel:[0,1,0] op_sel_hi:[1,1,1]\n\t"  // -n.y * d.y
        "v_pk_fma_f32 %[ms], %[E2], %[DMO], %[ms] op_sel:[0,1,0] op_sel_hi:[1,1,1]\n\t"    // E2' * MO'
        "v_pk_fma_f32 %[sh], %[e2z], %[moyz], %[sh] op_sel:[0,1,0] op_sel_hi:[1,1,1]\n\t"  // e2.z * mo.z
        "v_pk_fma_f32 %[det], %[nz], %[dzmx], %[det] op_sel:[0,0,0] op_sel_hi:[1,0,1]\n\t" // -n.z * d.z
        "v_pk_fma_f32 %[sh], %[mlx], %[dxy], %[sh] op_sel:[0,0,0] op_sel_hi:[1,0,1]\n\t"   // ml.x * d.x
        "v_pk_fma_f32 %[sh], %[mly], %[dxy], %[sh] op_sel:[0,1,0] op_sel_hi:[1,1,1]\n\t"   // ml.y * d.y
        "v_pk_fma_f32 %[sh], %[mlz], %[dzmx], %[sh] op_sel:[0,0,0] op_sel_hi:[1,0,1]"       // ml.z * d.z
        : [sh] "=&v"(sh), [det] "=&v"(det), [ms] "=&v"(ms), [md] "=&v"(md)
        : [e2x] "s"(tri_pair<0>(w)), [e2y] "s"(tri_pair<1>(w)), [e2z] "s"(tri_pair<2>(w)), [mlx] "s"(tri_pair<3>(w)), [mly] "s"(tri_pair<4>(w)),
          [mlz] "s"(tri_pair<5>(w)), [nx] "s"(tri_pair<6>(w)), [ny] "s"(tri_pair<7>(w)), [nz] "s"(tri_pair<8>(w)), [E2] "s"(tri_pair<9>(w)),
          [ML] "s"(tri_pair<10>(w)), [NN] "s"(tri_pair<11>(w)),
          [dxy] "v"(f.d_xy), [dzmx] "v"(f.dz_mox), [moyz] "v"(f.mo_yz), [DMO] "v"(f.D_MO));
    rejA = tri_filter_one(sh.x, det.x, ms.x, md.x);
    rejB = tri_filter_one(sh.y, det.y, ms.y, md.y);
}

// ---- strict pruning --------------------------------------------------------------------------------------------------
// The reference descends into every box that passes its WEAK test (three independent per-axis interval tests,
// bounding_box.rs:27-46): a ray without a hit yet enters every box that lies ahead of it on each axis separately, i.e.
// most of the tree.  The result, however, is decided by the triangles the reference ACCEPTS, and the walk may skip any
// subtree that holds none of those (the remaining visits happen in the same order with the same running best, so every
// accept, every tie and the final hit are unchanged -- DESIGN.md section 5, "Strict pruning", has the argument in full).
// A subtree is skipped when the ray's LINE certainly misses the box of the subtree's primitives (triangles, spheres,
// moving spheres; taken in the primitives' own space, so no transform enters the argument) inflated by tau:
//   * accepted triangle T, ray (o, d) in T's space  =>  the real point o + T^ d (T^ the exact solution the computed
//     u, v, t approximate) lies within tau = tau_coef (|o|_inf + radius) of T, PROVIDED the ray is not nearly
//     parallel to T's plane: |d.n^_T| >= 2^-25 |d|_inf.  (Nearly parallel rays make det tiny against its rounding
//     error and the reference can then accept garbage; mesh.rs:59 only rejects |det| < 2.2e-16.)
//   * ray_certified() establishes the proviso for EVERY triangle of the pruned space at once: a direction grid lists,
//     per cell, the (clustered) unit normals whose perpendicular band meets the cell; the ray checks those in f64.
//     Rays that fail (about 1 in 2000 on the pagoda) are walked without pruning, as are NaN-capable rays, rays out of
//     the fp32 range, instance sub-walks and the counting kernel -- which evaluates the prune decision anyway and
//     counts every accept inside a subtree that pruning would have skipped (cnt.bad, must stay 0).
// Per axis k the filter evaluates near'_k = min over the two planes of fma32(b', inv', c-) and far'_k = max of
// fma32(b', inv', c+) with c-/+ = RN32 towards -/+inf of (-o_k inv_k -/+ a_k), a_k = (tau + 2^-22 (B + |o_k|)) |inv_k|
// (1 + 2^-20): the inflation tau and every fp32 rounding (2^-24 (2 |b inv| + |o inv| + a_k), B >= |b|) are inside
// a_k, so near'_k <= the real entry parameter of the inflated slab and far'_k >= the real exit.  An accepted triangle
// implies max_k near'_k <= T^ <= min_k far'_k; the subtree is skipped iff max_k near'_k > min_k far'_k.
struct StrictFilter { f2 inv_xy, invz_cnx, cn_yz, cp_xy, cpz_; };
RT_DEV bool strict_filter_init(StrictFilter& f, d3 o, d3 d, const SceneDev& sc) {  // false: this lane must not prune
    const d3 inv = mk(1.0 / d.x, 1.0 / d.y, 1.0 / d.z);
    const double cx = -(o.x * inv.x), cy = -(o.y * inv.y), cz = -(o.z * inv.z);
    const double ax = fabs(inv.x), ay = fabs(inv.y), az = fabs(inv.z), ox = fabs(o.x), oy = fabs(o.y), oz = fabs(o.z);
    const double O = fmax(fmax(ox, oy), oz), B = sc.prune_box_abs_max, I = fmax(fmax(ax, ay), az);
    const double tau = sc.prune_tau_coef * (O + sc.prune_radius);
    const double mx = (tau + 0x1p-22 * (B + ox)) * ax * (1.0 + 0x1p-20), my = (tau + 0x1p-22 * (B + oy)) * ay * (1.0 + 0x1p-20),
                 mz = (tau + 0x1p-22 * (B + oz)) * az * (1.0 + 0x1p-20);
    const bool ok = fmin(fmin(ax, ay), az) >= 1e-30 && I <= 1e30 && O <= 1e30 && B >= 1e-30 && B <= 1e30 && (B + O) * I <= 1e36 &&
                    fmax(fmax(mx, my), mz) <= 1e30;   // also false for any NaN
    f.inv_xy = (f2){(float)inv.x, (float)inv.y};
    f.invz_cnx = (f2){(float)inv.z, f32_below(cx - mx)};
    f.cn_yz = (f2){f32_below(cy - my), f32_below(cz - mz)};
    f.cp_xy = (f2){f32_above(cx + mx), f32_above(cy + my)};
    f.cpz_ = (f2){f32_above(cz + mz), 0.f};
    return ok;
}
template <int B>  // B: first dword of the record's LOCAL box inside the window; returns the lanes whose line misses it
RT_DEV lanemask strict_filter(const StrictFilter& f, const Window& w) {
    f2 nx, ny, nz, fx, fy, fz;
    PK_FMA_BB(nx, wpair<B + 0>(w), f.inv_xy, 0, f.invz_cnx, 1);
    PK_FMA_BB(ny, wpair<B + 2>(w), f.inv_xy, 1, f.cn_yz, 0);
    PK_FMA_BB(nz, wpair<B + 4>(w), f.invz_cnx, 0, f.cn_yz, 1);
    PK_FMA_BB(fx, wpair<B + 0>(w), f.inv_xy, 0, f.cp_xy, 0);
    PK_FMA_BB(fy, wpair<B + 2>(w), f.inv_xy, 1, f.cp_xy, 1);
    PK_FMA_BB(fz, wpair<B + 4>(w), f.invz_cnx, 0, f.cpz_, 0);
    float n0, n1, n2, f0, f1, f2v;
    lanemask m;
    asm("v_min_f32 %[n0], %[nxa], %[nxb]\n\tv_max_f32 %[f0], %[fxa], %[fxb]\n\t"
        "v_min_f32 %[n1], %[nya], %[nyb]\n\tv_max_f32 %[f1], %[fya], %[fyb]\n\t"
        "v_min_f32 %[n2], %[nza], %[nzb]\n\tv_max_f32 %[f2], %[fza], %[fzb]\n\t"
        "v_max3_f32 %[n0], %[n0], %[n1], %[n2]\n\t"
        "v_min3_f32 %[f0], %[f0], %[f1], %[f2]\n\t"
        "v_cmp_gt_f32_e64 %[m], %[n0], %[f0]"
        : [n0] "=&v"(n0), [n1] "=&v"(n1), [n2] "=&v"(n2), [f0] "=&v"(f0), [f1] "=&v"(f1), [f2] "=&v"(f2v), [m] "=&s"(m)
        : [nxa] "v"(nx.x), [nxb] "v"(nx.y), [nya] "v"(ny.x), [nyb] "v"(ny.y), [nza] "v"(nz.x), [nzb] "v"(nz.y),
          [fxa] "v"(fx.x), [fxb] "v"(fx.y), [fya] "v"(fy.x), [fyb] "v"(fy.y), [fza] "v"(fz.x), [fzb] "v"(fz.y));
    return m;
}
// Is the direction (in the pruned space) at least 2^-24 |d|_inf away from perpendicular to every triangle normal of that
// space?  Cell of the folded cube map, then an exact f64 check of the normals listed for it (host side: cert_raster).
#ifndef RT_CERT_K
#define RT_CERT_K 1024u
#endif
#define RT_CERT_EPS 0x1p-24
struct CertEntry { float nx, ny, nz; uint32_t cluster; };  // a listed normal (fp32 copy for the first look) and its cluster
// A ray that is nearly parallel to a triangle's plane AND starts within delta of that plane is "suspect" for the
// triangle's leaf: the reference may accept garbage there, so subtrees holding that leaf are never pruned for this ray.
#define RT_SUSPECT_SLOTS 3
struct Suspects {
    uint32_t leaf[RT_SUSPECT_SLOTS];  // stream offsets of the suspect leaves, 0xFFFFFFFF = unused
    bool overflow;                    // more suspect leaves than slots: the ray is walked without pruning
    RT_DEV void add(uint32_t off) {
        bool placed = false;
#pragma unroll
        for (int k = 0; k < RT_SUSPECT_SLOTS; ++k) {
            if (placed || leaf[k] == off) { placed = true; continue; }
            if (leaf[k] == 0xFFFFFFFFu) { leaf[k] = off; placed = true; }
        }
        overflow = overflow || !placed;
    }
};
// second stage for one listed cluster: exact direction test, then the distance of the origin from the cluster's planes
RT_DEV void cluster_suspects(const SceneDev& sc, const uint32_t* __restrict__ node_off, uint32_t c, d3 o, d3 d, double thr, double delta, Suspects& sus) {
    const double* __restrict__ n = sc.cert_normals + 3u * (size_t)c;
    if (fabs(d.x * n[0] + d.y * n[1] + d.z * n[2]) >= thr) return;
    // Nearly parallel to the planes of this cluster.  The reference can only accept such a triangle if the origin
    // lies within delta of its plane (otherwise one of the u / v numerators exceeds det): collect the leaves of such planes.
    const double q = o.x * n[0] + o.y * n[1] + o.z * n[2];
    const double* __restrict__ pl = sc.cert_planes;
    const uint32_t* __restrict__ leaf = sc.cert_plane_leaf;
    uint32_t lo = sc.cert_plane_off[c], hi = sc.cert_plane_off[c + 1];
    const uint32_t last = hi;
    while (lo < hi) {  // first plane offset >= q - delta
        const uint32_t mid = (lo + hi) >> 1;
        if (pl[mid] < q - delta) lo = mid + 1; else hi = mid;
    }
    for (; lo < last && !(pl[lo] > q + delta) && !sus.overflow; ++lo) sus.add(node_off[leaf[lo]]);  // the leaf's record in the stream being walked
}
// false: the ray must be walked without pruning
RT_DEV bool ray_certified(const SceneDev& sc, const uint32_t* __restrict__ node_off, d3 o, d3 d, Suspects& sus) {
#pragma unroll
    for (int k = 0; k < RT_SUSPECT_SLOTS; ++k) sus.leaf[k] = 0xFFFFFFFFu;
    sus.overflow = false;
    const double ax = fabs(d.x), ay = fabs(d.y), az = fabs(d.z);
    const double D = fmax(fmax(ax, ay), az), O = fmax(fmax(fabs(o.x), fabs(o.y)), fabs(o.z));
    if (!(D >= 0x1p-100 && D <= 0x1p100 && O <= 0x1p100)) return false;  // also NaN
    if (sc.cert_n_normals == 0u) return true;  // a pruned space of spheres only: nothing can be nearly coplanar
    const int a = ax >= ay ? (ax >= az ? 0 : 2) : (ay >= az ? 1 : 2);
    const double da = a == 0 ? d.x : a == 1 ? d.y : d.z, db = a == 0 ? d.y : a == 1 ? d.z : d.x, dc = a == 0 ? d.z : a == 1 ? d.x : d.y;
    const double u = db / da, v = dc / da;
    const double kk = (double)RT_CERT_K * 0.5, top = (double)(RT_CERT_K - 1u);
    const uint32_t iu = (uint32_t)fmin(fmax(floor((u + 1.0) * kk), 0.0), top), iv = (uint32_t)fmin(fmax(floor((v + 1.0) * kk), 0.0), top);
    const uint32_t cell = ((uint32_t)a * RT_CERT_K + iv) * RT_CERT_K + iu;
    const uint32_t* __restrict__ cells = sc.cert_cells;
    const float4* __restrict__ entries = (const float4*)sc.cert_entries;
    const uint32_t beg = cells[cell], end = cells[cell + 1];  // lists are padded to a multiple of four entries
    const double thr = RT_CERT_EPS * D, delta = sc.prune_delta_coef * (O + sc.prune_radius);
    // first look in fp32 on the direction scaled to |d|_inf = 1: |error| <= 14 * 2^-24 (three products of factors <= 1, each
    // operand and operation rounded), so a value above 2^-24 + 2^-20 certifies the entry without the f64 test
    const float fx = (float)(d.x / D), fy = (float)(d.y / D), fz = (float)(d.z / D);
    const float thr32 = 0x1p-24f + 0x1p-20f;
    for (uint32_t e = beg; e < end; e += 4u) {
        const float4 e0 = entries[e], e1 = entries[e + 1u], e2 = entries[e + 2u], e3 = entries[e + 3u];
        const float m0 = fabsf(fx * e0.x + fy * e0.y + fz * e0.z), m1 = fabsf(fx * e1.x + fy * e1.y + fz * e1.z);
        const float m2 = fabsf(fx * e2.x + fy * e2.y + fz * e2.z), m3 = fabsf(fx * e3.x + fy * e3.y + fz * e3.z);
        if (!(fminf(fminf(m0, m1), fminf(m2, m3)) >= thr32)) {
            if (!(m0 >= thr32)) cluster_suspects(sc, node_off, __builtin_bit_cast(uint32_t, e0.w), o, d, thr, delta, sus);
            if (!(m1 >= thr32)) cluster_suspects(sc, node_off, __builtin_bit_cast(uint32_t, e1.w), o, d, thr, delta, sus);
            if (!(m2 >= thr32)) cluster_suspects(sc, node_off, __builtin_bit_cast(uint32_t, e2.w), o, d, thr, delta, sus);
            if (!(m3 >= thr32)) cluster_suspects(sc, node_off, __builtin_bit_cast(uint32_t, e3.w), o, d, thr, delta, sus);
        }
    }
    return !sus.overflow;
}
// everything box_step needs to prune: the local-box filter, the lanes allowed to prune, their suspect leaves
struct PruneState {
    StrictFilter sf;
    lanemask enable;
    uint32_t s0, s1, s2;
    bool any_suspect;  // wave-uniform
};
RT_DEV lanemask mask_lt_u32(uint32_t v, uint32_t s) { lanemask m; asm("v_cmp_lt_u32_e64 %0, %1, %2" : "=s"(m) : "v"(v), "s"(s)); return m; }

struct RayS { d3 o, d; double time; };  // ray as handed to this traversal level
// Where a traversal level gets its f64 ray from.  Only the fp32 filter state is live across the hot loop; the exact
// paths (a few percent of the steps) fetch the f64 ray again instead of holding 14 VGPRs per lane for it:
// the top level parks it in LDS (14 KB per block; volatile reads stay inside the rare blocks), an instance
// level keeps the transformed ray it was handed in registers.
// SHADOW flavours (MIS visibility rays, integrator.rs:82-86): the caller only needs to know whether the closest hit lies within
// 1e-4 of t_light = light.hit(ray).t, so a lane stops as soon as it ACCEPTS a hit with RN(t - t_light) < -1e-4: up to that
// accept its walk is the reference's own (same tests, same running best), the reference's final t can only be smaller, and
// rounding is monotone -- |t_final - t_light| > 1e-4 whatever comes after.  tl() is that ray's t_light.
template <bool SHADOW_, int STRIDE = 256> struct RayFromLdsT {   // STRIDE: threads per block (the LDS arrays are [component][thread])
    static constexpr bool SHADOW = SHADOW_;
    lptr<volatile double> slot;  // &lds[threadIdx.x]; component c at slot[c * STRIDE]; 7..9 = 1/d (bounding_box.rs:29), computed once per ray; 10 = t_light (SHADOW)
    RT_DEV double tl() const { return slot[10 * STRIDE]; }
    RT_DEV RayS get() const {
        RayS r;
        r.o = mk(slot[0 * STRIDE], slot[1 * STRIDE], slot[2 * STRIDE]);
        r.d = mk(slot[3 * STRIDE], slot[4 * STRIDE], slot[5 * STRIDE]);
        r.time = slot[6 * STRIDE];
        return r;
    }
    RT_DEV d3 inv() const { return mk(slot[7 * STRIDE], slot[8 * STRIDE], slot[9 * STRIDE]); }
};
typedef RayFromLdsT<false> RayFromLds;
template <bool SHADOW_> struct RayInRegistersT {
    static constexpr bool SHADOW = SHADOW_;
    RayS r;
    d3 rinv;
    double t_light;
    RT_DEV double tl() const { return t_light; }
    RT_DEV RayS get() const { return r; }
    RT_DEV d3 inv() const { return rinv; }
};
typedef RayInRegistersT<false> RayInRegisters;

template <int LEVEL, bool COUNT, bool INST, class RAYSRC, class HIT>
RT_DEV void traverse(const SceneDev& sc, uint32_t first, const RAYSRC& src, HIT& h, bool active, double t_min, uint32_t inst_prim, LaneCount& cnt);

RT_DEV bool ray_is_plain(d3 o, d3 inv) {  // box_axis_fast precondition: no NaN can appear in (b - o) * inv
    return isfinite(o.x) && isfinite(o.y) && isfinite(o.z) && isfinite(inv.x) && isfinite(inv.y) && isfinite(inv.z) && inv.x != 0.0 && inv.y != 0.0 && inv.z != 0.0;
}

// transform.rs:25-32: the ray in the space of transform `xform` (RT_NONE: the ray itself)
// PLAIN (every ray of the wave has finite coordinates and finite non-zero direction components) and xform == sc.translate_xf (the
// pagoda wraps every triangle in from_translation(5,5,5), main.rs:815-818): the matrix product ((1 x + 0 y) + 0 z) + t w rounds to
// x + t for the origin (the zeros add exactly, t != 0 absorbs the sign of a zero sum) and to x itself for the direction (x != 0,
// w = 0), so the shortcut below is the reference's result bit for bit at 3 additions instead of 21 multiplications and 18 additions.
template <bool PLAIN = false>
RT_DEV void apply_xform(const SceneDev& sc, uint32_t xform, const RayS& ray, d3& xo, d3& xd) {
    if (PLAIN && xform != RT_NONE && xform == sc.translate_xf) {
        cptr<double> m = as_const(sc.xforms[xform].t_inv);
        xo = mk(ray.o.x + m[12], ray.o.y + m[13], ray.o.z + m[14]); xd = ray.d;
    } else if (xform != RT_NONE) {
        cptr<double> m = as_const(sc.xforms[xform].t_inv);
        double mm[16];
#pragma unroll
        for (int k = 0; k < 16; ++k) mm[k] = m[k];
        xo = mat_mul_xyz(mm, ray.o, 1.0); xd = mat_mul_xyz(mm, ray.d, 0.0);
    } else {
        xo = ray.o; xd = ray.d;
    }
}

// One primitive against the wave's rays (exact f64); (xo, xd) is the ray already in the primitive's space.
// Returns true if the lane's closest hit changed.
template <int LEVEL, bool COUNT, bool INST, class HIT>
RT_DEV bool test_prim(const SceneDev& sc, uint32_t p, const GeomS& g, bool pass, const RayS& ray, HIT& h, d3 xo, d3 xd,
                      double t_min, uint32_t inst_prim, LaneCount& cnt) {
    bool changed = false;
    if (g.kind == RT_PRIM_TRIANGLE) {
        const d3 v0 = mk(g.d[0], g.d[1], g.d[2]), e1 = mk(g.d[3], g.d[4], g.d[5]), e2 = mk(g.d[6], g.d[7], g.d[8]);
        TriStage st;
        if (COUNT) cnt.tri += pass ? 1u : 0u;
        const bool alive = tri_stage_a(v0, e1, e2, xo, xd, st) & pass;
        if (__builtin_amdgcn_ballot_w64(alive) != 0ull) {  // rare: most tested triangles are nowhere near the rays
            double t;
            if (alive && tri_hit_tail(st, v0, e1, e2, xo, xd, t_min, h.best, t)) { h.best = t; h.prim = p; h.inst = inst_prim; changed = true; }
        }
    } else if (g.kind == RT_PRIM_SPHERE || g.kind == RT_PRIM_MOVING_SPHERE) {
        if (COUNT) cnt.sph += pass ? 1u : 0u;
        d3 c = mk(g.d[0], g.d[1], g.d[2]);
        double radius = g.d[3], t;
        if (g.kind == RT_PRIM_MOVING_SPHERE) { c = moving_center(c, mk(g.d[3], g.d[4], g.d[5]), g.d[6], g.d[7], ray.time); radius = g.d[8]; }
        if (pass && sphere_hit(c, radius, xo, xd, t_min, h.best, t)) { h.best = t; h.prim = p; h.inst = inst_prim; changed = true; }
    } else if (g.kind == RT_PRIM_AARECT) {
        double t;
        if (pass && rect_hit(g.d[0], g.d[1], g.d[2], g.d[3], g.d[4], xo, xd, t_min, h.best, t)) { h.best = t; h.prim = p; h.inst = inst_prim; changed = true; }
    } else if constexpr (LEVEL == 0 && INST) {  // RT_PRIM_INSTANCE: BVHNode object (optionally under a Transformed), main.rs:161-188
        RayInRegisters sub; sub.r.o = xo; sub.r.d = xd; sub.r.time = ray.time; sub.rinv = mk(1.0 / xd.x, 1.0 / xd.y, 1.0 / xd.z);
        const double before = h.best;
        traverse<1, COUNT, INST, RayInRegisters, HIT>(sc, (uint32_t)g.d[0], sub, h, pass, t_min, p, cnt);  // first node of the sub-array
        changed = (double)h.best != before;
    }
    return changed;
}

// Wave-coherent walk of the stream records in byte range [first, end).
//   * `off` is the wave-uniform cursor; a lane's ray takes part in a record iff its private `wait` is <= off.
//   * A ray that fails the box of record j sets wait = skip(j); because subtrees nest, any ray that is
//     NOT active at record i has wait >= skip(i).  Hence the wave goes to the next record if any ray
//     passed and to skip(i) otherwise -- no cross-lane reduction is needed.
//   * The rays are kept in the space of ONE transform (cur_xf) across the inner loop, where they are loop
//     invariant: transform.rs:25-32 runs once per (ray, run of leaves under the same matrix) -- the pagoda
//     wraps every triangle in the same translation.  A leaf whose first primitive wants another transform
//     leaves the inner loop; the outer loop re-derives the rays and re-enters at that record (its box test
//     is simply repeated: rays that failed it have already moved on, rays that passed pass again).
// MODE selects the box-test form:
//   BOX_LITERAL  bounding_box.rs:27-46 as written, in f64, for NaN-capable rays / boxes
//   BOX_PLAIN    fp32 filter, with the min/max f64 form (exact when no NaN can appear) for the unknown lanes
// (An octant-uniform form -- near/far planes picked by scalar selects -- measured slower: the scalar unit is
// as loaded as the vector unit in this loop.)
enum { BOX_LITERAL = 0, BOX_PLAIN = 1 };
#ifndef RT_LEAF_FIRST
#define RT_LEAF_FIRST 1
#endif
#ifndef RT_STRICT_FIRST
#define RT_STRICT_FIRST 1
#endif
#define RT_WALK_DONE 0xFFFFFFFEu  /* `want` value that ends the walk (never a transform id: RT_NONE is 0xFFFFFFFF, ids are < n_xforms) */
template <bool PLAIN>
RT_DEV bool box_exact(cptr<rt_node> nodes, uint32_t rec, d3 o, d3 inv, double t_min, double best) {
    if (rec & RT_NODE_NO_BOX) return true;  // hittable.rs:41-53: list items are tested without a box (also the END record, which has no node)
    const RT_CONST rt_node& nd = nodes[rec >> 8];
    if (PLAIN)
        return box_axis_fast(nd.bmin[0], nd.bmax[0], o.x, inv.x, t_min, best) & box_axis_fast(nd.bmin[1], nd.bmax[1], o.y, inv.y, t_min, best) &
               box_axis_fast(nd.bmin[2], nd.bmax[2], o.z, inv.z, t_min, best);
    return box_axis(nd.bmin[0], nd.bmax[0], o.x, inv.x, t_min, best) & box_axis(nd.bmin[1], nd.bmax[1], o.y, inv.y, t_min, best) &
           box_axis(nd.bmin[2], nd.bmax[2], o.z, inv.z, t_min, best);
}

// One record's box test for the wave: returns the mask of lanes that descend and moves the others to skip(record).
// `strict` (wave-uniform): the record is an inner one, whose local box may prune (StrictFilter).
template <int B, int MODE, bool COUNT, class RAYSRC, class HIT>
RT_DEV lanemask box_step(const Window& w, const BoxFilter& bf, const PruneState& ps, bool strict, uint32_t off, uint32_t& wait, uint32_t rec,
                         cptr<rt_node> nodes, const RAYSRC& src, HIT& h, double t_min, LaneCount& cnt) {
    const lanemask m_act = mask_reached(off, wait);
#ifdef RT_TRACE_DIAG
    cnt.lane_steps += mask_lane(m_act) ? 1u : 0u;
#endif
    if (COUNT) cnt.box += (mask_lane(m_act) && !(rec & RT_NODE_NO_BOX)) ? 1u : 0u;
    lanemask m_pass;
    if (MODE == BOX_PLAIN) {
        lanemask cp = 0ull, cf = 0ull, sp = 0ull;
        if (strict && ps.enable != 0ull && wword<B + 8>(w) != 0xFF800000u) {  // a local box starting at -inf is the (-inf, +inf) box of a subtree that is never pruned
            sp = strict_filter<B + 8>(ps.sf, w) & ps.enable;
            if (ps.any_suspect) {  // a subtree that holds one of the ray's suspect leaves is not pruned
                const uint32_t span = wword<B + 6>(w) - off;
                sp &= ~(mask_lt_u32(ps.s0 - off, span) | mask_lt_u32(ps.s1 - off, span) | mask_lt_u32(ps.s2 - off, span));
            }
        }
        const lanemask sp_apply = COUNT ? 0ull : sp;  // the counting kernel walks the reference's own visited set
#if RT_STRICT_FIRST
        if (!COUNT && (m_act & ~sp_apply) == 0ull) {  // every ray that is here is pruned: the reference's own test decides nothing
            wait = mask_select(wait, wword<B + 6>(w), m_act);
            return 0ull;
        }
#endif
        box_filter<B>(bf, w, cp, cf);
        m_pass = m_act & cp & ~sp_apply;
        if (COUNT || (m_act & ~(cp | cf | sp_apply)) != 0ull) {  // rare: a plane value within the error bound of best / t_min
#ifdef RT_TRACE_DIAG
            const uint64_t cx0 = __builtin_readcyclecounter();
#endif
            const RayS ray = src.get();
            const bool act = mask_lane(m_act);
            const bool ex = act & box_exact<true>(nodes, rec, ray.o, src.inv(), t_min, h.best);
            if (COUNT) {
                cnt.bad += (act & ((mask_lane(cp) & !ex) | (mask_lane(cf) & ex))) ? 1u : 0u;
                if (ex && mask_lane(sp) && off >= cnt.pruned_until) cnt.pruned_until = wword<B + 6>(w);  // pruning would skip up to here
            }
            m_pass = __builtin_amdgcn_ballot_w64(ex) & ~sp_apply;
#ifdef RT_TRACE_DIAG
            cnt.exact_box += 1; cnt.cyc_xbox += __builtin_readcyclecounter() - cx0;
#endif
        }
    } else {
        const RayS ray = src.get();
        m_pass = __builtin_amdgcn_ballot_w64(mask_lane(m_act) & box_exact<false>(nodes, rec, ray.o, src.inv(), t_min, h.best));
    }
    wait = mask_select(wait, wword<B + 6>(w), m_act & ~m_pass);
    return m_pass;
}

// Exact f64 tests of a TRI2 leaf for the lanes of m_pass the filter did not reject (COUNT: for every lane of m_pass),
// one triangle at a time (register pressure); the f64 ray is re-derived here rather than kept live across the hot loop.
template <bool COUNT, int MODE, class RAYSRC, class HIT>
RT_DEV bool tri2_exact(const SceneDev& sc, uint32_t node, uint32_t rec_off, lanemask m_pass, lanemask rejA, lanemask rejB, const RAYSRC& src, uint32_t cur_xf, HIT& h,
                       double t_min, uint32_t inst_prim, BoxFilter& bf, LaneCount& cnt) {
    cptr<rt_node> nodes = as_const(sc.nodes);
    cptr<rt_geom> geom = as_const(sc.geom);
#ifdef RT_TRACE_DIAG
    const uint64_t ct0 = __builtin_readcyclecounter();
#endif
    const RayS ray = src.get();
    d3 xo, xd;
    apply_xform<MODE == BOX_PLAIN>(sc, cur_xf, ray, xo, xd);
    const bool pass = mask_lane(m_pass);
    bool changed = false;
#pragma unroll
    for (int k = 0; k < 2; ++k) {
        const lanemask m_rej = k == 0 ? rejA : rejB;
        if (!COUNT && (m_pass & ~m_rej) == 0ull) continue;  // nobody needs this triangle
        const uint32_t p = k == 0 ? nodes[node].prim0 : nodes[node].prim1;
        if (p == RT_NONE) break;  // single-triangle leaf: its filter block repeats triangle A
        const GeomS g = load_geom(geom, p);
        const d3 v0 = mk(g.d[0], g.d[1], g.d[2]), e1 = mk(g.d[3], g.d[4], g.d[5]), e2 = mk(g.d[6], g.d[7], g.d[8]);
        const bool rej = mask_lane(m_rej);
        TriStage st;
        double t;
        const bool go = pass & (COUNT ? true : !rej) & tri_stage_a(v0, e1, e2, xo, xd, st);
        if (go && tri_hit_tail(st, v0, e1, e2, xo, xd, t_min, h.best, t)) {
            h.best = t; h.prim = p; h.inst = inst_prim; changed = true;
            // a filtered lane must be one the reference rejects, and an accept must not lie in a subtree pruning would skip
            if (COUNT) cnt.bad += (rej || rec_off < cnt.pruned_until) ? 1u : 0u;
        }
    }
    if (MODE == BOX_PLAIN && changed) box_filter_set_best(bf, h.best);
#ifdef RT_TRACE_DIAG
    cnt.exact_tri += 1; cnt.cyc_xtri += __builtin_readcyclecounter() - ct0;
#endif
    return changed;
}

template <int LEVEL, bool COUNT, bool INST, int MODE, class RAYSRC, class HIT>
RT_DEV void traverse_impl(const SceneDev& sc, uint32_t first, const RAYSRC& src, HIT& h, bool active, double t_min,
                          uint32_t inst_prim, LaneCount& cnt) {
    // which stream: the production form, or the one with a record for every node (counting kernel, NaN-capable waves: the
    // nesting argument behind the production form needs NaN-free slab values).  `first`: LEVEL 0 -- unused (the scene range
    // starts at root_off), LEVEL 1 -- first node of the instance's sub-array.
    const bool full = COUNT || MODE == BOX_LITERAL;
    constexpr bool SHADOW = RAYSRC::SHADOW && LEVEL == 0 && !COUNT;  // the counting kernel walks the reference's complete visited set
    const uint32_t end_off = full ? sc.end_off_full : sc.end_off;    // the END record of the scene range
    // A lane whose accepted hit already decides "not visible" waits for the END record; if that was the last lane the walk is over,
    // otherwise the cursor jumps to the first record any lane still needs.  Only ever reached from wave-uniform control flow.
#define RT_SHADOW_FINISH(CHANGED)                                                                  \
    if constexpr (SHADOW) {                                                                        \
        const bool ch_ = (CHANGED);                                                                \
        if (__builtin_amdgcn_ballot_w64(ch_) != 0ull) {                                            \
            const bool fin_ = ch_ && ((double)h.best - src.tl()) < -0.0001;                        \
            if (__builtin_amdgcn_ballot_w64(fin_) != 0ull) {                                       \
                wait = fin_ ? end_off : wait;                                                      \
                const uint32_t nxt_ = wave_min_u32(wait);                                          \
                if (nxt_ >= end_off) { want = RT_WALK_DONE; break; }                               \
                off = nxt_ > off ? nxt_ : off;                                                     \
            }                                                                                      \
        }                                                                                          \
        RT_RECONVERGE();                                                                           \
    }
    const uint32_t* node_off = full ? sc.node_off_full : sc.node_off;
    first = LEVEL == 0 ? (full ? sc.root_off_full : sc.root_off) : as_const(node_off)[first];
    uint32_t wait = active ? first : 0xFFFFFFFFu;
    // strict pruning: top-level walks of NaN-free rays in scenes that have a pruned space; the walk starts in that space
    const bool prune = LEVEL == 0 && MODE == BOX_PLAIN && sc.prune_on != 0u;
    uint32_t off = first, cur_xf = prune ? sc.prune_xf : RT_NONE;
    lanemask clean = 0ull;
    PruneState ps;
    ps.enable = 0ull; ps.s0 = ps.s1 = ps.s2 = 0xFFFFFFFFu; ps.any_suspect = false;
    if (prune) {
        const RayS ray = src.get();
        d3 xo, xd;
        apply_xform<MODE == BOX_PLAIN>(sc, sc.prune_xf, ray, xo, xd);
        Suspects sus;
        const bool c = ray_certified(sc, node_off, xo, xd, sus) && active && ray.time >= sc.prune_time0 && ray.time <= sc.prune_time1;
        if (COUNT) cnt.uncertified += (active && !c) ? 1u : 0u;
#ifdef RT_TRACE_DIAG
        cnt.diag_flags |= (active && !c) ? 2u : 0u;
        cnt.diag_flags |= (active && sus.leaf[0] != 0xFFFFFFFFu) ? 8u : 0u;
#endif
        clean = __builtin_amdgcn_ballot_w64(c);
        ps.s0 = sus.leaf[0]; ps.s1 = sus.leaf[1]; ps.s2 = sus.leaf[2];
        ps.any_suspect = __builtin_amdgcn_ballot_w64(c && sus.leaf[0] != 0xFFFFFFFFu) != 0ull;
        RT_RECONVERGE();
    }
    const uint64_t base = (uint64_t)(full ? sc.stream_full : sc.stream);
    cptr<rt_node> nodes = as_const(sc.nodes);
    cptr<rt_geom> geom = as_const(sc.geom);
    BoxFilter bf;
    if (MODE == BOX_PLAIN) {
        const RayS ray = src.get();
        box_filter_init(bf, ray.o, src.inv(), t_min, h.best, sc.box_abs_max);
    }
    for (;;) {  // one round per transform the walk has to enter; the END record of the range leaves both loops
        TriFilter tf;
        ps.enable = 0ull;
        {
            const RayS ray = src.get();
            d3 xo, xd;
            apply_xform<MODE == BOX_PLAIN>(sc, cur_xf, ray, xo, xd);
            tri_filter_init(tf, xo, xd);
            if (prune && cur_xf == sc.prune_xf) {  // inner records carry boxes of THIS space only
                const bool ok = strict_filter_init(ps.sf, xo, xd, sc);
                ps.enable = clean & __builtin_amdgcn_ballot_w64(ok);
#ifdef RT_TRACE_DIAG
                cnt.diag_flags |= (active && !ok) ? 4u : 0u;
#endif
            } else {
                ps.sf.inv_xy = ps.sf.invz_cnx = ps.sf.cn_yz = ps.sf.cp_xy = ps.sf.cpz_ = (f2){0.f, 0.f};
            }
        }
        uint32_t want = cur_xf;
        for (;;) {
            Window w;
#ifdef RT_TRACE_DIAG
            const uint64_t c0 = __builtin_readcyclecounter();
#endif
            off = (uint32_t)__builtin_amdgcn_readfirstlane((int)off);  // the nested walk / the per-lane certificate loop make the compiler class the (wave-uniform) cursor divergent
            window_issue(w, base, off);
            window_wait(w);
            const uint32_t skip = wword<6>(w), rec = wword<7>(w);
            const uint32_t node = rec >> 8;
            cnt.wsteps += 1u;
#ifdef RT_TRACE_DIAG
            cnt.cyc_wait += __builtin_readcyclecounter() - c0; cnt.steps += 1; cnt.leaf_steps += (rec & RT_STREAM_KIND) ? 1u : 0u;
            if (cnt.cyc_setup == 0) cnt.cyc_setup = c0 - cnt.cyc_total;   // kernel entry (cyc_total holds the start stamp) to the first record
            if (cnt.leaf_open) cnt.cyc_leaf += c0 - cnt.leaf_open;
            cnt.leaf_open = (rec & RT_STREAM_KIND) ? c0 : 0;
#endif
            if (RT_LEAF_FIRST && MODE == BOX_PLAIN && !COUNT && (rec & RT_STREAM_KIND) == RT_STREAM_TRI2 && wword<8>(w) == cur_xf) {
                // Common leaf, rays already in its space: triangle filter FIRST.  The leaf's box only gates its own
                // triangle tests (skip == next record either way), so it is evaluated for a wave only when some ray is
                // not certainly rejected by both triangles -- the weak box test of the reference passes most leaves
                // of a ray that has no hit yet, so the box would rarely have saved the triangle filter anyway.
                const lanemask m_act = mask_reached(off, wait);
                lanemask rejA, rejB;
                tri_filter(tf, w, rejA, rejB);
                bool changed = false;
                if ((m_act & ~(rejA & rejB)) != 0ull) {
                    const lanemask m_pass = box_step<0, MODE, COUNT>(w, bf, ps, false, off, wait, rec, nodes, src, h, t_min, cnt);
                    if ((m_pass & ~(rejA & rejB)) != 0ull) changed = tri2_exact<COUNT, MODE>(sc, node, off, m_pass, rejA, rejB, src, cur_xf, h, t_min, inst_prim, bf, cnt);
                    RT_RECONVERGE();
                }
                off += RT_STREAM_REC_TRI2;
                RT_SHADOW_FINISH(changed)
                continue;
            }
            lanemask m_pass = box_step<0, MODE, COUNT>(w, bf, ps, (rec & RT_STREAM_KIND) == RT_STREAM_INNER, off, wait, rec, nodes, src, h, t_min, cnt);
            if (m_pass == 0ull) {
                off = skip;
            } else if ((rec & RT_STREAM_KIND) == RT_STREAM_INNER) {
                off += RT_STREAM_REC_INNER;
                // the window already holds the next record's header: if that is an inner node too, take it in the same step
                // (no load, no wait, no trip round the loop) -- a quarter of all steps on the pagoda before strict pruning
                const uint32_t rec2 = wword<23>(w);
                if ((rec2 & RT_STREAM_KIND) == RT_STREAM_INNER) {
#ifdef RT_TRACE_DIAG
                    cnt.steps += 1;
#endif
                    cnt.wsteps += 1u;
                    m_pass = box_step<16, MODE, COUNT>(w, bf, ps, true, off, wait, rec2, nodes, src, h, t_min, cnt);
                    off = m_pass == 0ull ? wword<22>(w) : off + RT_STREAM_REC_INNER;
                }
            } else if ((rec & RT_STREAM_KIND) == RT_STREAM_SPH2) {  // one or two spheres: exact f64 test (shape.rs:93-121) on inline geometry
                // the spheres' own transform is applied on the spot: only the triangle filter needs the walk to BE in a space
                off += RT_STREAM_REC_SPH2;
                const uint32_t nsph = wword<9>(w);
                const RayS ray = src.get();
                d3 xo, xd;
                apply_xform<MODE == BOX_PLAIN>(sc, wword<8>(w), ray, xo, xd);
                const bool pass = mask_lane(m_pass);
                if (COUNT) cnt.sph += pass ? (nsph + ((rec & RT_NODE_DUP_RIGHT) ? 1u : 0u)) : 0u;
                double t;
                bool changed = false;
                if (pass && sphere_hit(mk(sdbl(wword<10>(w), wword<11>(w)), sdbl(wword<12>(w), wword<13>(w)), sdbl(wword<14>(w), wword<15>(w))),
                                       sdbl(wword<16>(w), wword<17>(w)), xo, xd, t_min, h.best, t)) {
                    h.best = t; h.prim = nodes[node].prim0; h.inst = inst_prim; changed = true;
                }
                if (nsph > 1u) {
                    if (pass && sphere_hit(mk(sdbl(wword<18>(w), wword<19>(w)), sdbl(wword<20>(w), wword<21>(w)), sdbl(wword<22>(w), wword<23>(w))),
                                           sdbl(wword<24>(w), wword<25>(w)), xo, xd, t_min, h.best, t)) {
                        h.best = t; h.prim = nodes[node].prim1; h.inst = inst_prim; changed = true;
                    }
                }
                if (MODE == BOX_PLAIN && changed) box_filter_set_best(bf, h.best);
                RT_RECONVERGE();
                RT_SHADOW_FINISH(changed)
            } else if ((rec & RT_STREAM_KIND) == RT_STREAM_TRI2) {  // the common leaf: one or two triangles under one transform
                if (wword<8>(w) != cur_xf) {  // re-enter at this record with the rays in the leaf's space
                    if (COUNT && !(rec & RT_NODE_NO_BOX)) cnt.box -= mask_lane(m_pass) ? 1u : 0u;  // the repeated box test must not count twice
                    want = wword<8>(w);
                    break;
                }
                off += RT_STREAM_REC_TRI2;
                if (COUNT) {  // reference-order test count: two triangles, or one (tested twice as the duplicated child of a 1-object BVH leaf)
                    const uint32_t ntests = (nodes[node].prim1 != RT_NONE || (rec & RT_NODE_DUP_RIGHT)) ? 2u : 1u;
                    cnt.tri += mask_lane(m_pass) ? ntests : 0u;
                }
                lanemask rejA, rejB;
                tri_filter(tf, w, rejA, rejB);
                bool changed = false;
                if (COUNT || (m_pass & ~(rejA & rejB)) != 0ull) {  // rare: most tested triangles are nowhere near the rays
                    changed = tri2_exact<COUNT, MODE>(sc, node, off - RT_STREAM_REC_TRI2, m_pass, rejA, rejB, src, cur_xf, h, t_min, inst_prim, bf, cnt);
                    RT_RECONVERGE();
                }
                RT_SHADOW_FINISH(changed)
            } else {  // any other leaf: exact tests on the blob's tables
                if (node == RT_STREAM_END_NODE) { want = RT_WALK_DONE; break; }  // END record: every ray still waiting has arrived at the end of the range
                const uint32_t p0 = wword<8>(w), p1 = wword<9>(w);
                const bool pass = mask_lane(m_pass);
                const uint32_t x0 = wword<10>(w);  // transform of the first primitive, applied on the spot
                off += RT_STREAM_REC_LEAF;
                const GeomS g0 = load_geom(geom, p0);
                const RayS ray = src.get();
                d3 xo, xd;
                apply_xform<MODE == BOX_PLAIN>(sc, x0, ray, xo, xd);
                bool changed = test_prim<LEVEL, COUNT, INST, HIT>(sc, p0, g0, pass, ray, h, xo, xd, t_min, inst_prim, cnt);
                if (p1 != RT_NONE) {
                    const GeomS g1 = load_geom(geom, p1);
                    if (g1.xform == x0) {
                        changed |= test_prim<LEVEL, COUNT, INST, HIT>(sc, p1, g1, pass, ray, h, xo, xd, t_min, inst_prim, cnt);
                    } else {  // second primitive of a leaf under a different matrix: one-off transform
                        d3 yo, yd;
                        apply_xform<MODE == BOX_PLAIN>(sc, g1.xform, ray, yo, yd);
                        changed |= test_prim<LEVEL, COUNT, INST, HIT>(sc, p1, g1, pass, ray, h, yo, yd, t_min, inst_prim, cnt);
                    }
                }
                if (COUNT && (rec & RT_NODE_DUP_RIGHT)) {  // reference-order count of the skipped duplicate test
                    if (g0.kind == RT_PRIM_TRIANGLE) cnt.tri += pass ? 1u : 0u;
                    else if (g0.kind == RT_PRIM_SPHERE || g0.kind == RT_PRIM_MOVING_SPHERE) cnt.sph += pass ? 1u : 0u;
                }
                if (MODE == BOX_PLAIN && changed) box_filter_set_best(bf, h.best);
                RT_RECONVERGE();
                RT_SHADOW_FINISH(changed)
            }
        }
#undef RT_SHADOW_FINISH
        if (want == RT_WALK_DONE) break;
        cur_xf = (uint32_t)__builtin_amdgcn_readfirstlane((int)want);
    }
}

template <int LEVEL, bool COUNT, bool INST, class RAYSRC, class HIT>
RT_DEV void traverse(const SceneDev& sc, uint32_t first, const RAYSRC& src, HIT& h, bool active, double t_min, uint32_t inst_prim, LaneCount& cnt) {
    bool odd;
    {
        const RayS ray = src.get();
        const d3 inv = src.inv();  // bounding_box.rs:29
        odd = active && !(ray_is_plain(ray.o, inv) && t_min <= h.best);
    }
    // the min/max form of the slab test is exact unless a NaN can appear (zero / non-finite direction or origin): decided per wave
    const bool plain = sc.boxes_finite && __builtin_amdgcn_ballot_w64(odd) == 0ull;
    if (plain) traverse_impl<LEVEL, COUNT, INST, BOX_PLAIN, RAYSRC, HIT>(sc, first, src, h, active, t_min, inst_prim, cnt);
    else {
#ifdef RT_TRACE_DIAG
        if (LEVEL == 0) cnt.diag_flags |= odd ? 1u : 0u;
#endif
        traverse_impl<LEVEL, COUNT, INST, BOX_LITERAL, RAYSRC, HIT>(sc, first, src, h, active, t_min, inst_prim, cnt);
    }
}

#ifndef RT_TRACE_BLOCK
#define RT_TRACE_BLOCK 64    /* threads per k_trace block: ONE wave.  A block keeps its LDS until its last wave ends, and the walks are long-tailed (mean 260 steps,
                                * max 5 600): with four waves per block the finished waves' LDS stayed allocated and kept new blocks out.  256 / 128 / 64: 1 960 / 2 018 / 2 090 Mrays/s (r02k) */
#endif
#ifndef RT_TRACE_WAVES
#define RT_TRACE_WAVES 6  /* waves per SIMD the register allocator must leave room for */
#endif
#ifndef RT_TRACE_WAVES_REGS
#define RT_TRACE_WAVES_REGS 3
#endif
// rays: one 64-byte block per path (ox,oy,oz,dx | dy,dz,time,pad); outputs indexed by path.
// REGS: keep the f64 ray in registers instead of LDS -- for scenes whose leaves are mostly not triangles, where
// (almost) every leaf step runs an exact f64 test and would read the ray back from LDS each time.
// SHADOW: the rays are MIS visibility rays whose block carries t_light in its last slot; see RayFromLdsT.
template <bool COUNT, bool INST, bool REGS, bool SHADOW = false>
__global__ __launch_bounds__(RT_TRACE_BLOCK) __attribute__((amdgpu_waves_per_eu(INST ? 2 : REGS ? RT_TRACE_WAVES_REGS : RT_TRACE_WAVES))) void k_trace(SceneDev sc, const uint32_t* __restrict__ queue, const uint32_t* __restrict__ n_ptr, uint32_t n_fixed,
                                               const double* __restrict__ rays, uint32_t cap, double t_min, double t_max,
                                               uint32_t* __restrict__ out_prim, uint32_t* __restrict__ out_inst, double* __restrict__ out_t,
                                               unsigned long long* __restrict__ stats, uint32_t lanes, uint32_t t_stride, PathHit* __restrict__ out_hit,
                                               const uint32_t* __restrict__ remap) {
    // `lanes` rays per wave (64, or fewer for incoherent queues: the wave walks the UNION of its rays' visited records)
    // remap (k_walk's left-overs): the launch covers the n queue SLOTS listed in remap instead of slots 0..n-1
    const uint32_t n = n_ptr ? *n_ptr : n_fixed;
    const uint32_t tid = blockIdx.x * blockDim.x + threadIdx.x;
    const uint32_t wave_first = (uint32_t)__builtin_amdgcn_readfirstlane((int)((tid >> 6) * lanes));
    if (tid == 0 && stats && !remap) atomicAdd(&stats[0], (unsigned long long)n);  // rays traced (remapped rays were counted by k_walk_setup)
    if (wave_first >= n) return;
    const bool valid = (tid & 63u) < lanes && wave_first + (tid & 63u) < n;
    const uint32_t slot = remap ? (valid ? remap[wave_first + (tid & 63u)] : 0u) : wave_first + (tid & 63u);
    const uint32_t path = valid ? (queue ? queue[slot] : slot) : 0u;
    __shared__ double lds_ray[(SHADOW ? 11 : 10) * RT_TRACE_BLOCK];
    __shared__ double lds_best[RT_TRACE_BLOCK];
    __shared__ uint32_t lds_prim[RT_TRACE_BLOCK], lds_inst[RT_TRACE_BLOCK];
    {
        double* slot = &lds_ray[threadIdx.x];
        const double4* r4 = (const double4*)(rays + (size_t)path * RT_RAY_STRIDE);
        const double4 a = valid ? r4[0] : make_double4(0.0, 0.0, 0.0, 1.0), b = valid ? r4[1] : make_double4(1.0, 1.0, 0.0, 0.0);
        slot[0 * RT_TRACE_BLOCK] = a.x; slot[1 * RT_TRACE_BLOCK] = a.y; slot[2 * RT_TRACE_BLOCK] = a.z; slot[3 * RT_TRACE_BLOCK] = a.w; slot[4 * RT_TRACE_BLOCK] = b.x; slot[5 * RT_TRACE_BLOCK] = b.y; slot[6 * RT_TRACE_BLOCK] = b.z;
        slot[7 * RT_TRACE_BLOCK] = 1.0 / a.w; slot[8 * RT_TRACE_BLOCK] = 1.0 / b.x; slot[9 * RT_TRACE_BLOCK] = 1.0 / b.y;
        if (SHADOW) slot[10 * RT_TRACE_BLOCK] = b.w;  // t_light (k_shade)
    }
    RayFromLdsT<SHADOW, RT_TRACE_BLOCK> src; src.slot = (lptr<volatile double>)&lds_ray[threadIdx.x];  // each lane reads back only its own slot: no barrier needed
    RayInRegistersT<SHADOW> src_regs;
    if (REGS) { src_regs.r = src.get(); src_regs.rinv = src.inv(); src_regs.t_light = SHADOW ? src.tl() : 0.0; }
    LaneHit h;
    h.best.p = (lptr<volatile double>)&lds_best[threadIdx.x]; h.prim.p = (lptr<volatile uint32_t>)&lds_prim[threadIdx.x]; h.inst.p = (lptr<volatile uint32_t>)&lds_inst[threadIdx.x];
    h.best = t_max; h.prim = RT_NONE; h.inst = RT_NONE;
    LaneCount cnt; cnt.box = cnt.tri = cnt.sph = cnt.bad = 0; cnt.pruned_until = 0; cnt.uncertified = 0; cnt.wsteps = 0;
#ifdef RT_TRACE_DIAG
    cnt.steps = cnt.leaf_steps = cnt.cyc_wait = cnt.lane_steps = cnt.exact_box = cnt.exact_tri = 0; cnt.cyc_xbox = cnt.cyc_xtri = cnt.cyc_setup = cnt.cyc_leaf = cnt.leaf_open = 0; cnt.cyc_total = __builtin_readcyclecounter(); cnt.diag_flags = 0;
#endif
    if (REGS) {
        LaneHitRegs hr; hr.best = t_max; hr.prim = RT_NONE; hr.inst = RT_NONE;
        traverse<0, COUNT, INST, RayInRegistersT<SHADOW>, LaneHitRegs>(sc, sc.root_off, src_regs, hr, valid, t_min, RT_NONE, cnt);
        h.best = (double)hr.best; h.prim = (uint32_t)hr.prim; h.inst = (uint32_t)hr.inst;
    } else {
        traverse<0, COUNT, INST, RayFromLdsT<SHADOW, RT_TRACE_BLOCK>, LaneHit>(sc, sc.root_off, src, h, valid, t_min, RT_NONE, cnt);
    }
#ifdef RT_TRACE_DIAG
    cnt.cyc_total = __builtin_readcyclecounter() - cnt.cyc_total;
    if (cnt.steps > 5000 && (cnt.diag_flags != 0u || lane_id() == 0)) {
        const RayS r = src.get();
        printf("[diag-slow] steps %llu lane %u flags %u o %.17g %.17g %.17g d %.17g %.17g %.17g hit %u\n", (unsigned long long)cnt.steps, lane_id(), cnt.diag_flags,
               r.o.x, r.o.y, r.o.z, r.d.x, r.d.y, r.d.z, (uint32_t)h.prim);
    }
    const uint64_t lane_steps = wave_sum_u64(cnt.lane_steps);
    if (stats && lane_id() == 0) {
        atomicAdd(&stats[9], (unsigned long long)lane_steps);
        atomicMax(&stats[10], (unsigned long long)cnt.steps);
        atomicAdd(&stats[11], (unsigned long long)cnt.exact_box);
        atomicAdd(&stats[12], (unsigned long long)cnt.exact_tri);
        atomicAdd(&stats[4], (unsigned long long)cnt.steps); atomicAdd(&stats[5], (unsigned long long)cnt.leaf_steps);
        atomicAdd(&stats[6], (unsigned long long)cnt.cyc_wait); atomicAdd(&stats[7], (unsigned long long)cnt.cyc_total); atomicAdd(&stats[8], 1ull);
        if (!COUNT) {   // cycle shares (the counting kernel uses these slots for its test counts)
            atomicAdd(&stats[13], (unsigned long long)cnt.cyc_xtri); atomicAdd(&stats[14], (unsigned long long)cnt.cyc_xbox);
            atomicAdd(&stats[1], (unsigned long long)cnt.cyc_setup); atomicAdd(&stats[2], (unsigned long long)cnt.cyc_leaf);
        }
    }
#endif
    if (sc.count_steps != 0u && stats && lane_id() == 0) atomicAdd(&stats[15], (unsigned long long)(uint32_t)__builtin_amdgcn_readfirstlane((int)cnt.wsteps));
    if (valid) {
        const uint32_t hp = h.prim;
        const double ht = hp != RT_NONE ? (double)h.best : __builtin_nan("");
        if (out_hit) {  // the render loop: one 16-byte block per QUEUE SLOT (k_shade walks the same queue: both sides coalesced)
            PathHit ph; ph.t = ht; ph.prim = hp; ph.inst = (uint32_t)h.inst;
            out_hit[slot] = ph;
        } else {
            if (out_prim) out_prim[path] = hp;
            if (out_inst) out_inst[path] = (uint32_t)h.inst;
            out_t[(size_t)path * t_stride] = ht;
        }
    }
    if (COUNT && stats) {
        uint64_t b = wave_sum_u64(cnt.box), t = wave_sum_u64(cnt.tri), s = wave_sum_u64(cnt.sph), bad = wave_sum_u64(cnt.bad);
        const uint64_t unc = wave_sum_u64(cnt.uncertified);
        if (lane_id() == 0) {
            atomicAdd(&stats[14], (unsigned long long)unc);
            atomicAdd(&stats[1], (unsigned long long)b); atomicAdd(&stats[2], (unsigned long long)t); atomicAdd(&stats[3], (unsigned long long)s);
            atomicAdd(&stats[13], (unsigned long long)bad);
        }
    }
}

// ------------------------------------------------------------------------------------------------
// k_walk: PER-LANE walk of the same trace stream, for the incoherent launches (bounce >= 1, shadow rays)
// ------------------------------------------------------------------------------------------------
// k_trace charges a wave for the UNION of its 64 rays' records: secondary rays share little of their walks, so a wave of
// them steps through 260-850 records with 21-35 % of its lanes active (profiles/r02b_diag.txt).  Here every lane keeps its OWN
// cursor and fetches its record with vector loads, and a lane whose ray is finished takes the next ray of the queue
// (one atomic per wave), so lanes stay busy whatever their neighbours visit.  What is decided and how is unchanged:
// the same stream, the same fp32 filters (weak box, triangle, strict / local box with certificate) with the same
// constants, the same exact f64 fallbacks on the blob's tables for undecided lanes -- a lane visits its records in the
// reference's order with the reference's running best, so (primitive, t) is bit-identical to k_trace's.
// Work is split three ways so that no block of code runs for a handful of lanes:
//   RUN   one record per lane: header + local box (64 B), weak and strict filter, cursor update          (fp32, ~4/5 of the steps)
//   TRI   a TRI2 leaf: 112-B triangle block, triangle filter, then the leaf's weak box filter            (fp32)
//   SLOW  anything the filters could not decide or that needs f64: exact box, exact triangles, spheres, other leaves
// Each iteration the wave runs ONE of these for the lanes waiting for it (or fetches new rays), chosen by lane counts.
// MEASURED (r02): bit-identical frames, lane use 49 % instead of 21-35 %, and 1.6x SLOWER than k_trace
// on the pagoda's secondary / shadow launches -- every lane-step pulls its own 128-byte line through the CU's vector L1
// (64 B/clk), which is what bounds it (more waves per SIMD change nothing), whereas k_trace's one scalar fetch serves the
// whole wave.  It therefore stays OPT-IN (RT_RENDER_LANE_WALK) as the A/B counterpart; rt_render uses k_trace by default.
// k_walk_setup computes per ray, at full lane use, everything k_trace's set-up derives (1/d, filter constants, certificate)
// and stores it as 160 B of state; rays that need the literal NaN-capable box test are listed for a k_trace launch instead.
#define RT_WALK_STATE_VEC 10u   /* uint4 per ray, stored as [vec][cap] so that every load / store is coalesced */
#define RT_WS_ODD 1u            /* not a PLAIN ray (zero / non-finite component): left to k_trace's literal mode */
#ifndef RT_WALK_CHUNK
#define RT_WALK_CHUNK 512u   /* queue slots a wave claims per atomic */
#endif
#ifndef RT_WALK_FETCH_MIN
#define RT_WALK_FETCH_MIN 8
#endif
#ifndef RT_WALK_TRI_MIN
#define RT_WALK_TRI_MIN 20
#endif
#ifndef RT_WALK_SLOW_MIN
#define RT_WALK_SLOW_MIN 16
#endif
struct WalkRegs {   // the per-lane filter state (what BoxFilter / TriFilter / StrictFilter hold in k_trace), 32 dwords + 1/d
    float inv[3], c[3], M, hiP, hiF, loP, loF;   // weak box filter, WORLD space (node boxes are the reference's own)
    float d[3], mo[3], D, MO;                     // triangle filter, pruned space
    float sinv[3], cn[3], cp[3];                  // strict filter, pruned space; a ray that may not prune has cn = -inf, cp = +inf
    uint32_t s0, s1, s2, flags;                   // suspect leaves (stream offsets); RT_WS_*
};
union WalkPack { struct { WalkRegs r; double inv64[3]; uint32_t pad[2]; } s; uint4 v[RT_WALK_STATE_VEC]; };
static_assert(sizeof(WalkPack) == RT_WALK_STATE_VEC * 16, "walk state is 160 bytes");

__global__ __launch_bounds__(256) void k_walk_setup(SceneDev sc, const uint32_t* __restrict__ queue, const uint32_t* __restrict__ n_ptr, const double* __restrict__ rays,
                                                   uint4* __restrict__ state, uint32_t cap, double t_min, double t_max,
                                                   uint32_t* __restrict__ odd_list, uint32_t* __restrict__ odd_count, unsigned long long* __restrict__ stats) {
    const uint32_t n = *n_ptr;
    const uint32_t slot = blockIdx.x * blockDim.x + threadIdx.x;
    if (slot == 0 && stats) atomicAdd(&stats[0], (unsigned long long)n);  // rays traced
    if (slot >= n) return;
    const uint32_t path = queue[slot];
    const double4* r4 = (const double4*)(rays + (size_t)path * RT_RAY_STRIDE);
    const double4 a = r4[0], b = r4[1];
    RayS ray; ray.o = mk(a.x, a.y, a.z); ray.d = mk(a.w, b.x, b.y); ray.time = b.z;
    const d3 inv = mk(1.0 / ray.d.x, 1.0 / ray.d.y, 1.0 / ray.d.z);   // bounding_box.rs:29
    WalkPack P;
    WalkRegs& R = P.s.r;
    P.s.inv64[0] = inv.x; P.s.inv64[1] = inv.y; P.s.inv64[2] = inv.z; P.s.pad[0] = P.s.pad[1] = 0u;
    const bool plain = sc.boxes_finite && ray_is_plain(ray.o, inv) && t_min <= t_max;
    R.flags = plain ? 0u : RT_WS_ODD;
    if (!plain) odd_list[atomicAdd(odd_count, 1u)] = slot;
    BoxFilter bf;
    box_filter_init(bf, ray.o, inv, t_min, t_max, sc.box_abs_max);
    R.inv[0] = bf.inv_xy.x; R.inv[1] = bf.inv_xy.y; R.inv[2] = bf.invz_cx.x; R.c[0] = bf.invz_cx.y; R.c[1] = bf.c_yz.x; R.c[2] = bf.c_yz.y;
    R.M = bf.M; R.hiP = bf.hiP; R.hiF = bf.hiF; R.loP = bf.loP; R.loF = bf.loF;
    d3 xo, xd;
    apply_xform(sc, sc.prune_xf, ray, xo, xd);   // the pruned space (transform.rs:25-32)
    TriFilter tf;
    tri_filter_init(tf, xo, xd);
    R.d[0] = tf.d_xy.x; R.d[1] = tf.d_xy.y; R.d[2] = tf.dz_mox.x; R.mo[0] = tf.dz_mox.y; R.mo[1] = tf.mo_yz.x; R.mo[2] = tf.mo_yz.y; R.D = tf.D_MO.x; R.MO = tf.D_MO.y;
    Suspects sus;
    const bool certified = plain && ray_certified(sc, sc.node_off, xo, xd, sus) && ray.time >= sc.prune_time0 && ray.time <= sc.prune_time1;
    StrictFilter sf;
    const bool prune = strict_filter_init(sf, xo, xd, sc) && certified;
    const float inf = __builtin_inff();
    R.sinv[0] = sf.inv_xy.x; R.sinv[1] = sf.inv_xy.y; R.sinv[2] = sf.invz_cnx.x;
    R.cn[0] = prune ? sf.invz_cnx.y : -inf; R.cn[1] = prune ? sf.cn_yz.x : -inf; R.cn[2] = prune ? sf.cn_yz.y : -inf;
    R.cp[0] = prune ? sf.cp_xy.x : inf; R.cp[1] = prune ? sf.cp_xy.y : inf; R.cp[2] = prune ? sf.cpz_.x : inf;
    if (!prune) { R.sinv[0] = R.sinv[1] = R.sinv[2] = 0.f; }   // fma(b, 0, -+inf) = -+inf for every finite b: near' = -inf, far' = +inf, never "near' > far'"
    R.s0 = sus.leaf[0]; R.s1 = sus.leaf[1]; R.s2 = sus.leaf[2];
#pragma unroll
    for (uint32_t k = 0; k < RT_WALK_STATE_VEC; ++k) state[(size_t)k * cap + slot] = P.v[k];
}

// exact pieces with PER-LANE indices (k_trace's versions read wave-uniform records through scalar loads)
RT_DEV bool box_exact_lane(const rt_node* __restrict__ nodes, uint32_t rec, d3 o, d3 inv, double t_min, double best) {
    if (rec & RT_NODE_NO_BOX) return true;
    const rt_node& nd = nodes[rec >> 8];
    return box_axis_fast(nd.bmin[0], nd.bmax[0], o.x, inv.x, t_min, best) & box_axis_fast(nd.bmin[1], nd.bmax[1], o.y, inv.y, t_min, best) &
           box_axis_fast(nd.bmin[2], nd.bmax[2], o.z, inv.z, t_min, best);
}
RT_DEV void apply_xform_lane(const SceneDev& sc, uint32_t xform, const RayS& ray, d3& xo, d3& xd) {
    xo = ray.o; xd = ray.d;
    if (xform != RT_NONE) {
        const double* m = sc.xforms[xform].t_inv;
        double mm[16];
#pragma unroll
        for (int k = 0; k < 16; ++k) mm[k] = m[k];
        xo = mat_mul_xyz(mm, ray.o, 1.0); xd = mat_mul_xyz(mm, ray.d, 0.0);
    }
}
RT_DEV bool tri_filter_one_lane(float sh, float det, float ms, float md) {  // tri_filter_one with per-lane results: true = certainly rejected by the reference
    const float a = fabsf(sh) - ms, c = fabsf(det) - md;
    const float b = __builtin_fmaf(fabsf(det), 1.0f + 0x1p-20f, md);
    const float p = sh * det;
    return (a >= 2e-18f) & ((a > b) | ((c >= 2e-18f) & (0.f > p)));
}

enum { WST_IDLE = 0, WST_RUN = 1, WST_TRI = 2, WST_SLOW = 3 };
#define RT_WAUX_XBOX 1u   /* the weak box filter was undecided: exact f64 box test first */
#define RT_WAUX_REJA 2u   /* TRI2: the filter certainly rejected triangle A / B */
#define RT_WAUX_REJB 4u

template <bool SHADOW>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(4, 4))) void k_walk(SceneDev sc, const uint32_t* __restrict__ queue, const uint32_t* __restrict__ n_ptr,
        const double* __restrict__ rays, const uint4* __restrict__ state, uint32_t cap, double t_min, double t_max,
        PathHit* __restrict__ out_hit, double* __restrict__ out_t, uint32_t t_stride, uint32_t* __restrict__ next_slot, unsigned long long* __restrict__ stats) {
    const uint32_t n = *n_ptr;
    __shared__ double lds_ray[(SHADOW ? 11 : 10) * 256];
    __shared__ double lds_best[256];
    __shared__ uint32_t lds_prim[256];
    RayFromLdsT<SHADOW> src; src.slot = (lptr<volatile double>)&lds_ray[threadIdx.x];   // each lane only ever touches its own slots: no barrier
    lptr<volatile double> best_cell = (lptr<volatile double>)&lds_best[threadIdx.x];
    lptr<volatile uint32_t> prim_cell = (lptr<volatile uint32_t>)&lds_prim[threadIdx.x];
    const uint8_t* __restrict__ base = sc.stream;
    const rt_node* __restrict__ nodes = sc.nodes;
    const rt_geom* __restrict__ geom = sc.geom;
    const uint32_t end_off = sc.end_off;
    const uint32_t lane = lane_id();
    const float ninf = -__builtin_inff();

    WalkRegs R;
    uint32_t st = WST_IDLE, off = 0u, slot = 0u, aux = 0u;
    bool exhausted = false;        // wave-uniform: the queue has no ray left
    uint32_t chunk_next = 0u, chunk_end = 0u;   // wave-uniform: the slots this wave has claimed and not yet handed to a lane
    bool any_suspect = false;      // wave-uniform: some lane's ray has a suspect leaf (1 ray in ~1e7)
    uint32_t n_run = 0, n_tri = 0, n_slow = 0, n_iter = 0;   // wave-uniform statistics: lane-steps per block, iterations
#define F(u) __builtin_bit_cast(float, u)

    for (;;) {
        const uint64_t m_idle = __builtin_amdgcn_ballot_w64(st == WST_IDLE), m_run = __builtin_amdgcn_ballot_w64(st == WST_RUN);
        const uint64_t m_tri = __builtin_amdgcn_ballot_w64(st == WST_TRI), m_slow = __builtin_amdgcn_ballot_w64(st == WST_SLOW);
        const uint32_t c_idle = (uint32_t)__popcll(m_idle), c_run = (uint32_t)__popcll(m_run), c_tri = (uint32_t)__popcll(m_tri), c_slow = (uint32_t)__popcll(m_slow);
        n_iter += 1u;
        if (!exhausted && c_idle != 0u && (c_idle >= RT_WALK_FETCH_MIN || (c_run | c_tri | c_slow) == 0u)) {
            // ---- FETCH: the idle lanes take the next rays of the wave's chunk; a wave claims RT_WALK_CHUNK queue slots with ONE atomic
            // (the counter is a single address: an atomic per fetch, 4 M per launch, ran at the rate the memory side serialises them)
            if (chunk_next >= chunk_end) {
                uint32_t got = 0u;
                if (lane == 0u) got = atomicAdd(next_slot, RT_WALK_CHUNK);
                got = (uint32_t)__builtin_amdgcn_readfirstlane((int)got);
                if (got >= n) { exhausted = true; any_suspect = __builtin_amdgcn_ballot_w64(st != WST_IDLE && R.s0 != 0xFFFFFFFFu) != 0ull; continue; }   // this wave asks no more
                chunk_next = got; chunk_end = got + RT_WALK_CHUNK < n ? got + RT_WALK_CHUNK : n;
            }
            const uint32_t first = chunk_next, avail = chunk_end - chunk_next;
            chunk_next += c_idle < avail ? c_idle : avail;
            if (st == WST_IDLE) {
                const uint32_t rank = (uint32_t)__popcll(m_idle & ((1ull << lane) - 1ull)), mine = first + rank;
                if (rank < avail) {
                    slot = mine;
                    WalkPack P;
#pragma unroll
                    for (uint32_t k = 0; k < RT_WALK_STATE_VEC; ++k) P.v[k] = state[(size_t)k * cap + slot];
                    R = P.s.r;
                    if (!(R.flags & RT_WS_ODD)) {   // ODD rays are traced by k_trace's literal mode (listed by k_walk_setup): the lane stays idle
                        const uint32_t path = queue[slot];
                        const double4* r4 = (const double4*)(rays + (size_t)path * RT_RAY_STRIDE);
                        const double4 a = r4[0], b = r4[1];
                        lptr<volatile double> s = src.slot;
                        s[0 * 256] = a.x; s[1 * 256] = a.y; s[2 * 256] = a.z; s[3 * 256] = a.w; s[4 * 256] = b.x; s[5 * 256] = b.y; s[6 * 256] = b.z;
                        s[7 * 256] = P.s.inv64[0]; s[8 * 256] = P.s.inv64[1]; s[9 * 256] = P.s.inv64[2];
                        if (SHADOW) s[10 * 256] = b.w;   // t_light
                        *best_cell = t_max; *prim_cell = RT_NONE;
                        off = sc.root_off; st = WST_RUN;
                    }
                }
            }
            any_suspect = __builtin_amdgcn_ballot_w64(st != WST_IDLE && R.s0 != 0xFFFFFFFFu) != 0ull;
            continue;
        }
        if ((c_run | c_tri | c_slow) == 0u) break;   // every lane idle and nothing left to fetch
        const bool do_slow = c_slow >= RT_WALK_SLOW_MIN || (c_run == 0u && c_slow >= c_tri);
        const bool do_tri = !do_slow && (c_tri >= RT_WALK_TRI_MIN || c_run == 0u);
        bool finished = false;     // this lane's ray is done: write its result below
        if (do_slow) {
            n_slow += c_slow;
            if (st == WST_SLOW) {
                // ---- SLOW: the reference's f64 arithmetic for what the filters left open
                const uint2 sr = *(const uint2*)(base + off + 24u);
                const uint32_t skip = sr.x, rec = sr.y, kind = rec & RT_STREAM_KIND, node = rec >> 8, rec_bytes = ((rec >> 4) & 15u) * 16u;
                const RayS ray = src.get();
                double best = *best_cell;
                bool pass = true;
                if (aux & RT_WAUX_XBOX) pass = box_exact_lane(nodes, rec, ray.o, src.inv(), t_min, best);
                if (kind == RT_STREAM_INNER) {
                    off = pass ? off + RT_STREAM_REC_INNER : skip;
                } else {
                    if (pass) {
                        bool changed = false;
                        uint32_t hit_prim = RT_NONE;
                        if (kind == RT_STREAM_TRI2) {   // mesh.rs:48-81 on the leaf's triangles the filter did not reject, A then B
                            d3 xo, xd;
                            apply_xform(sc, sc.prune_xf, ray, xo, xd);
#pragma unroll
                            for (int k = 0; k < 2; ++k) {
                                const uint32_t p = k == 0 ? nodes[node].prim0 : nodes[node].prim1;
                                if (p == RT_NONE || (aux & (k == 0 ? RT_WAUX_REJA : RT_WAUX_REJB))) continue;   // single-triangle leaf: its block repeats A
                                const rt_geom& g = geom[p];
                                const d3 v0 = mk(g.d[0], g.d[1], g.d[2]), e1 = mk(g.d[3], g.d[4], g.d[5]), e2 = mk(g.d[6], g.d[7], g.d[8]);
                                TriStage ts;
                                double t;
                                if (tri_stage_a(v0, e1, e2, xo, xd, ts) && tri_hit_tail(ts, v0, e1, e2, xo, xd, t_min, best, t)) { best = t; hit_prim = p; changed = true; }
                            }
                        } else if (kind == RT_STREAM_SPH2) {   // shape.rs:93-121 on the inline f64 geometry
                            const uint32_t xf = *(const uint32_t*)(base + off + 32u), nsph = *(const uint32_t*)(base + off + 36u);
                            const double* sp = (const double*)(base + off + 40u);
                            d3 xo, xd;
                            apply_xform_lane(sc, xf, ray, xo, xd);
                            double t;
                            if (sphere_hit(mk(sp[0], sp[1], sp[2]), sp[3], xo, xd, t_min, best, t)) { best = t; hit_prim = nodes[node].prim0; changed = true; }
                            if (nsph > 1u && sphere_hit(mk(sp[4], sp[5], sp[6]), sp[7], xo, xd, t_min, best, t)) { best = t; hit_prim = nodes[node].prim1; changed = true; }
                        } else {   // any other leaf: the blob's own tables (no instances in scenes this kernel serves)
                            const uint32_t p0 = *(const uint32_t*)(base + off + 32u), p1 = *(const uint32_t*)(base + off + 36u);
#pragma unroll
                            for (int k = 0; k < 2; ++k) {
                                const uint32_t p = k == 0 ? p0 : p1;
                                if (p == RT_NONE) continue;
                                const rt_geom& g = geom[p];
                                d3 xo, xd;
                                apply_xform_lane(sc, g.xform, ray, xo, xd);
                                double t;
                                bool got = false;
                                if (g.kind == RT_PRIM_TRIANGLE) {
                                    const d3 v0 = mk(g.d[0], g.d[1], g.d[2]), e1 = mk(g.d[3], g.d[4], g.d[5]), e2 = mk(g.d[6], g.d[7], g.d[8]);
                                    TriStage ts;
                                    got = tri_stage_a(v0, e1, e2, xo, xd, ts) && tri_hit_tail(ts, v0, e1, e2, xo, xd, t_min, best, t);
                                } else if (g.kind == RT_PRIM_SPHERE) {
                                    got = sphere_hit(mk(g.d[0], g.d[1], g.d[2]), g.d[3], xo, xd, t_min, best, t);
                                } else if (g.kind == RT_PRIM_MOVING_SPHERE) {
                                    got = sphere_hit(moving_center(mk(g.d[0], g.d[1], g.d[2]), mk(g.d[3], g.d[4], g.d[5]), g.d[6], g.d[7], ray.time), g.d[8], xo, xd, t_min, best, t);
                                } else if (g.kind == RT_PRIM_AARECT) {
                                    got = rect_hit(g.d[0], g.d[1], g.d[2], g.d[3], g.d[4], xo, xd, t_min, best, t);
                                }
                                if (got) { best = t; hit_prim = p; changed = true; }
                            }
                        }
                        if (changed) {
                            *best_cell = best; *prim_cell = hit_prim;
                            R.hiP = f32_below(best - (double)R.M); R.hiF = f32_above(best + (double)R.M);   // box_filter_set_best
                            if (SHADOW && (best - src.tl()) < -0.0001) finished = true;   // already decides "not visible" (see RayFromLdsT)
                        }
                    }
                    off += rec_bytes;
                }
                st = WST_RUN;
            }
        } else if (do_tri) {
            n_tri += c_tri;
            if (st == WST_TRI) {
                // ---- TRI: a TRI2 leaf (the rays are in its space: k_walk serves scenes whose triangle leaves all sit in the pruned space)
                const uint4* q = (const uint4*)(base + off);
                const uint4 h0 = q[0], h1 = q[1], b0 = q[2], b1 = q[3], b2 = q[4], b3 = q[5], b4 = q[6], b5 = q[7], b6 = q[8];
                // block: b0 = xform | pad | e2x(A,B); b1 = e2y, e2z; b2 = mlx, mly; b3 = mlz, nx; b4 = ny, nz; b5 = E2, ML; b6 = NN, -
                float shA = F(b0.z) * R.mo[0], shB = F(b0.w) * R.mo[0];
                float detA = F(b3.z) * R.d[0], detB = F(b3.w) * R.d[0];
                float msA = F(b5.z) * R.D, msB = F(b5.w) * R.D;
                const float mdA = F(b6.x) * R.D, mdB = F(b6.y) * R.D;
                shA = __builtin_fmaf(F(b1.x), R.mo[1], shA); shB = __builtin_fmaf(F(b1.y), R.mo[1], shB);
                detA = __builtin_fmaf(F(b4.x), R.d[1], detA); detB = __builtin_fmaf(F(b4.y), R.d[1], detB);
                msA = __builtin_fmaf(F(b5.x), R.MO, msA); msB = __builtin_fmaf(F(b5.y), R.MO, msB);
                shA = __builtin_fmaf(F(b1.z), R.mo[2], shA); shB = __builtin_fmaf(F(b1.w), R.mo[2], shB);
                detA = __builtin_fmaf(F(b4.z), R.d[2], detA); detB = __builtin_fmaf(F(b4.w), R.d[2], detB);
                shA = __builtin_fmaf(F(b2.x), R.d[0], shA); shB = __builtin_fmaf(F(b2.y), R.d[0], shB);
                shA = __builtin_fmaf(F(b2.z), R.d[1], shA); shB = __builtin_fmaf(F(b2.w), R.d[1], shB);
                shA = __builtin_fmaf(F(b3.x), R.d[2], shA); shB = __builtin_fmaf(F(b3.y), R.d[2], shB);
                const bool rejA = tri_filter_one_lane(shA, detA, msA, mdA), rejB = tri_filter_one_lane(shB, detB, msB, mdB);
                if (rejA & rejB) {   // the leaf's box gates nothing but its own triangles
                    off += RT_STREAM_REC_TRI2; st = WST_RUN;
                } else {
                    const float tx0 = __builtin_fmaf(F(h0.x), R.inv[0], R.c[0]), tx1 = __builtin_fmaf(F(h0.y), R.inv[0], R.c[0]);
                    const float ty0 = __builtin_fmaf(F(h0.z), R.inv[1], R.c[1]), ty1 = __builtin_fmaf(F(h0.w), R.inv[1], R.c[1]);
                    const float tz0 = __builtin_fmaf(F(h1.x), R.inv[2], R.c[2]), tz1 = __builtin_fmaf(F(h1.y), R.inv[2], R.c[2]);
                    const float nearmax = vmax3_f32(vmin_f32(tx0, tx1), vmin_f32(ty0, ty1), vmin_f32(tz0, tz1));
                    const float farmin = vmin3_f32(vmax_f32(tx0, tx1), vmax_f32(ty0, ty1), vmax_f32(tz0, tz1));
                    const bool cp = (nearmax <= R.hiP) & (farmin >= R.loP), cf = (nearmax > R.hiF) | (farmin < R.loF);
                    if (cf) { off += RT_STREAM_REC_TRI2; st = WST_RUN; }
                    else { aux = (cp ? 0u : RT_WAUX_XBOX) | (rejA ? RT_WAUX_REJA : 0u) | (rejB ? RT_WAUX_REJB : 0u); st = WST_SLOW; }
                }
            }
        } else {
            n_run += c_run;
            if (st == WST_RUN) {
                // ---- RUN: one record: header (box | skip | flags) and, for an inner record, the local box of its subtree
                const uint4* q = (const uint4*)(base + off);
                const uint4 h0 = q[0], h1 = q[1], l0 = q[2];
                const uint2 l1 = *(const uint2*)(base + off + 48u);
                const uint32_t skip = h1.z, rec = h1.w, kind = rec & RT_STREAM_KIND;
                // weak filter: the reference's own box test, decided in fp32 where the error bound allows (BoxFilter)
                const float tx0 = __builtin_fmaf(F(h0.x), R.inv[0], R.c[0]), tx1 = __builtin_fmaf(F(h0.y), R.inv[0], R.c[0]);
                const float ty0 = __builtin_fmaf(F(h0.z), R.inv[1], R.c[1]), ty1 = __builtin_fmaf(F(h0.w), R.inv[1], R.c[1]);
                const float tz0 = __builtin_fmaf(F(h1.x), R.inv[2], R.c[2]), tz1 = __builtin_fmaf(F(h1.y), R.inv[2], R.c[2]);
                const float nearmax = vmax3_f32(vmin_f32(tx0, tx1), vmin_f32(ty0, ty1), vmin_f32(tz0, tz1));
                const float farmin = vmin3_f32(vmax_f32(tx0, tx1), vmax_f32(ty0, ty1), vmax_f32(tz0, tz1));
                const bool cp = (nearmax <= R.hiP) & (farmin >= R.loP), cf = (nearmax > R.hiF) | (farmin < R.loF);
                if (kind == RT_STREAM_INNER) {
                    // strict filter: the ray's line against the local box (StrictFilter); a box starting at -inf is never pruned
                    const float nx0 = __builtin_fmaf(F(l0.x), R.sinv[0], R.cn[0]), nx1 = __builtin_fmaf(F(l0.y), R.sinv[0], R.cn[0]);
                    const float ny0 = __builtin_fmaf(F(l0.z), R.sinv[1], R.cn[1]), ny1 = __builtin_fmaf(F(l0.w), R.sinv[1], R.cn[1]);
                    const float nz0 = __builtin_fmaf(F(l1.x), R.sinv[2], R.cn[2]), nz1 = __builtin_fmaf(F(l1.y), R.sinv[2], R.cn[2]);
                    const float fx0 = __builtin_fmaf(F(l0.x), R.sinv[0], R.cp[0]), fx1 = __builtin_fmaf(F(l0.y), R.sinv[0], R.cp[0]);
                    const float fy0 = __builtin_fmaf(F(l0.z), R.sinv[1], R.cp[1]), fy1 = __builtin_fmaf(F(l0.w), R.sinv[1], R.cp[1]);
                    const float fz0 = __builtin_fmaf(F(l1.x), R.sinv[2], R.cp[2]), fz1 = __builtin_fmaf(F(l1.y), R.sinv[2], R.cp[2]);
                    const float sn = vmax3_f32(vmin_f32(nx0, nx1), vmin_f32(ny0, ny1), vmin_f32(nz0, nz1));
                    const float sfar = vmin3_f32(vmax_f32(fx0, fx1), vmax_f32(fy0, fy1), vmax_f32(fz0, fz1));
                    bool pruned = (l0.x != 0xFF800000u) & (sn > sfar);
                    if (any_suspect) {   // a subtree that holds one of the ray's suspect leaves is not pruned
                        const uint32_t span = skip - off;
                        pruned = pruned & !((R.s0 - off < span) | (R.s1 - off < span) | (R.s2 - off < span));
                    }
                    if (pruned | cf) off = skip;
                    else if (cp) off += RT_STREAM_REC_INNER;
                    else { aux = RT_WAUX_XBOX; st = WST_SLOW; }
                } else if (kind == RT_STREAM_TRI2) {
                    st = WST_TRI;
                } else if (off == end_off) {   // the END record: every record this ray needed has been visited
                    finished = true;
                } else {   // sphere leaf / other leaf
                    if (cf) off += ((rec >> 4) & 15u) * 16u;
                    else { aux = cp ? 0u : RT_WAUX_XBOX; st = WST_SLOW; }
                }
            }
        }
        if (finished) {
            const uint32_t hp = *prim_cell;
            const double ht = hp != RT_NONE ? (double)*best_cell : __builtin_nan("");
            if (SHADOW) out_t[(size_t)queue[slot] * t_stride] = ht;
            else { PathHit ph; ph.t = ht; ph.prim = hp; ph.inst = RT_NONE; out_hit[slot] = ph; }
            st = WST_IDLE;
        }
    }
#undef F
    if (stats && lane == 0u) {
        atomicAdd(&stats[4], (unsigned long long)n_run); atomicAdd(&stats[5], (unsigned long long)n_tri);
        atomicAdd(&stats[6], (unsigned long long)n_slow); atomicAdd(&stats[7], (unsigned long long)n_iter);
    }
}

// ------------------------------------------------------------------------------------------------
// k_raygen: main.rs:996-999 + camera.rs:83-101
// ------------------------------------------------------------------------------------------------
// Slot -> pixel permutation of one sample layer: the 8-aligned part of the frame in 8x8 tiles (a wave of
// primary rays is then a square pixel block, whose rays share far more of the walk than a 64x1 strip),
// the right and bottom remainders row-major.  Results are keyed on the pixel, not the slot.
RT_DEV uint32_t tiled_pixel(uint32_t q, uint32_t width, uint32_t height) {
    const uint32_t w8 = width & ~7u, h8 = height & ~7u;
    if (q < w8 * h8) {
        const uint32_t tile = q >> 6, i = q & 63u, tpr = w8 >> 3;
        return ((tile / tpr) * 8u + (i >> 3)) * width + (tile % tpr) * 8u + (i & 7u);
    }
    uint32_t r = q - w8 * h8;
    const uint32_t rw = width - w8;
    if (r < rw * h8) return (r / rw) * width + w8 + r % rw;
    r -= rw * h8;
    return (h8 + r / width) * width + r % width;
}
__global__ __launch_bounds__(256) void k_raygen(rt_camera cam, uint32_t width, uint32_t height, uint32_t n, uint32_t sample0, uint32_t npix,
                                                const uint32_t* __restrict__ in_pixel, const uint32_t* __restrict__ in_sample, uint64_t seed,
                                                double* __restrict__ rays, uint32_t cap, PathInfo* __restrict__ info, uint32_t tiled) {
    const uint32_t p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= n) return;
    const uint32_t pixel = in_pixel ? in_pixel[p] : (tiled ? tiled_pixel(p % npix, width, height) : p % npix);
    const uint32_t sample = in_sample ? in_sample[p] : sample0 + p / npix;
    const uint32_t x = pixel % width, y = pixel / width;
    Rng rng(seed, pixel, sample, 0u);
    const double u = ((double)x + rng.next_f64()) / (double)(width - 1);
    const double v = ((double)y + rng.next_f64()) / (double)(height - 1);
    const d3 cu = ld3(cam.u), cv = ld3(cam.v), origin = ld3(cam.origin);
    d3 random_in_lens = rand_in_unit_sphere(rng) * cam.lens_radius;
    d3 offset = (cu * random_in_lens.x) + (cv * random_in_lens.y);
    double time = (cam.time1 - cam.time0) > 0.000001 ? rng.range(cam.time0, cam.time1) : 0.0;
    d3 ro = origin + offset;
    d3 rd = ld3(cam.lower_left_corner) + (ld3(cam.horizontal) * u) + (ld3(cam.vertical) * v) - origin - offset;
    double4* r4 = (double4*)(rays + (size_t)p * RT_RAY_STRIDE);
    r4[0] = make_double4(ro.x, ro.y, ro.z, rd.x); r4[1] = make_double4(rd.y, rd.z, time, 0.0);
    if (info) { PathInfo pi; pi.pixel = pixel; pi.sample = sample; pi.nb = 0u; pi.pad = 0u; info[p] = pi; }
}

// ------------------------------------------------------------------------------------------------
// k_shade: one ray_colour level (integrator.rs) for every live path
// ------------------------------------------------------------------------------------------------
// Bounce record, [depth][cap][REC_STRIDE] (a path's record is one 96-byte block: paths are shaded in sorted, not path, order).  With emitted == 0 for every scattering material:
//   not visible / no light sample:  L = (brdf*cos*L_next)/pdf              (integrator.rs:62-66,164-168,224,291,361)
//   light visible (MIS):            L = (brdf*cos*L_next*w)/pdf + C_l      (integrator.rs:138-154), w gated by L_next >= .001
enum { REC_BRDF_X = 0, REC_BRDF_Y, REC_BRDF_Z, REC_COS, REC_PDF, REC_W, REC_CL_X, REC_CL_Y, REC_CL_Z, REC_T_LIGHT, REC_T_SHADOW, REC_FIELDS, REC_STRIDE = 12 };

struct PathBuffers {
    double* rays;        // [7][cap]
    double* shadow_rays; // [7][cap]
    uint32_t* hit_prim; uint32_t* hit_inst; double* hit_t;  // rt_trace_closest only
    PathHit* hit;        // [cap] closest hit of the current ray, by slot of the current queue
    PathInfo* info;      // [cap] pixel, sample, records written
    double* terminal;    // [3][cap] radiance returned by the last ray_colour level
    double* records;     // [depth][cap][REC_STRIDE]
    double* radiance;    // [3][cap]
    unsigned long long* sig;  // [cap] path signatures (rt_sample_radiance_sig only; NULL otherwise)
    uint32_t cap;
};

// Coherence key of a secondary ray: direction octant (3 bits) | Morton code of the origin cell
// (5 bits per axis over the scene box).  The reference's per-axis slab test makes the visited node
// set a function of "which boxes lie ahead of the origin on each axis", i.e. of octant and origin:
// rays with equal keys walk nearly the same nodes, so a wave of them wastes few lanes.
RT_DEV uint32_t spread5(uint32_t v) { v &= 31u; v = (v | (v << 8)) & 0x100Fu; v = (v | (v << 4)) & 0x10C3u; v = (v | (v << 2)) & 0x1249u; return v; }
RT_DEV uint32_t sort_key(const SceneDev& sc, d3 o, d3 d) {
    const uint32_t oct = (d.x < 0.0 ? 1u : 0u) | (d.y < 0.0 ? 2u : 0u) | (d.z < 0.0 ? 4u : 0u);
    const float fx = ((float)o.x - sc.sort_min[0]) * sc.sort_scale[0], fy = ((float)o.y - sc.sort_min[1]) * sc.sort_scale[1], fz = ((float)o.z - sc.sort_min[2]) * sc.sort_scale[2];
    const uint32_t cx = (uint32_t)fminf(fmaxf(fx, 0.f), 31.f), cy = (uint32_t)fminf(fmaxf(fy, 0.f), 31.f), cz = (uint32_t)fminf(fmaxf(fz, 0.f), 31.f);
    // direction bin inside the octant (2 bits per axis of |d| / max|d|), least significant: rays leaving one cell in
    // similar directions tend to share their fate (first hit or none), which is what keeps a wave's lanes busy
    // after the first hits.  Measured against octant|cell alone: -2.5 % bench time; direction-major or finer
    // grids (6 bits per axis) were no better.
    const float ax = fabsf((float)d.x), ay = fabsf((float)d.y), az = fabsf((float)d.z);
    const float inv = 3.999f / fmaxf(fmaxf(ax, ay), fmaxf(az, 1e-30f));
    const uint32_t dbin = ((uint32_t)(ax * inv) & 3u) | (((uint32_t)(ay * inv) & 3u) << 2) | (((uint32_t)(az * inv) & 3u) << 4);
#ifndef RT_KEY_VARIANT
#define RT_KEY_VARIANT 2  /* measured with strict pruning, 16 spp: variant 2 -4 % trace time against 1, variants 3 and 4 no better than 1 */
#endif
    const uint32_t morton = spread5(cx) | (spread5(cy) << 1) | (spread5(cz) << 2);
#if RT_KEY_VARIANT == 1
    return (oct << 21) | (morton << 6) | dbin;
#elif RT_KEY_VARIANT == 2   // coarser cells (4 bits per axis), finer directions (3 bits per axis)
    const float inv8 = 7.999f / fmaxf(fmaxf(ax, ay), fmaxf(az, 1e-30f));
    const uint32_t d9 = ((uint32_t)(ax * inv8) & 7u) | (((uint32_t)(ay * inv8) & 7u) << 3) | (((uint32_t)(az * inv8) & 7u) << 6);
    const uint32_t m4 = spread5(cx >> 1) | (spread5(cy >> 1) << 1) | (spread5(cz >> 1) << 2);
    return (oct << 21) | ((m4 & 0xFFFu) << 9) | d9;
#elif RT_KEY_VARIANT == 3   // origin-major
    return (morton << 9) | (oct << 6) | dbin;
#elif RT_KEY_VARIANT == 4   // cell, then direction, Morton-interleaved pairs: 3 origin bits, then octant, then the rest
    return ((morton >> 9) << 18) | (oct << 15) | ((morton & 0x1FFu) << 6) | dbin;
#endif
}
#define RT_SORT_BITS 24u  /* octant (3) | origin cell (12) | direction bin (9): three 8-bit digit passes */

RT_DEV void push_queue_keyed(bool want, uint32_t path, uint32_t key, uint32_t* __restrict__ queue, uint32_t* __restrict__ keys, uint32_t* __restrict__ count) {
    const uint64_t m = __ballot(want);
    if (m == 0ull) return;
    uint32_t base = 0;
    const uint32_t lane = lane_id();
    const uint32_t leader = (uint32_t)__builtin_ctzll(m);
    if (lane == leader) base = atomicAdd(count, (uint32_t)__popcll(m));
    base = (uint32_t)__shfl((int)base, (int)leader, 64);
    if (want) {
        const uint32_t slot = base + (uint32_t)__popcll(m & ((1ull << lane) - 1ull));
        queue[slot] = path;
        if (keys) keys[slot] = key;
    }
}

// k_shade extends its two queues with ONE atomic per block and queue (after RT_SHADE_TILES tiles of 256 slots), not one per wave: the
// counters are single addresses, and device-scope atomics on one address from 8 XCDs serialise at several ns each -- 6 M of them per
// 50-spp pass were a third of k_shade's time (r02g: one per 256 slots 1 837 -> 1 965 Mrays/s, same frame).
#ifndef RT_SHADE_BLOCK
#define RT_SHADE_BLOCK 256u   /* threads per k_shade block */
#endif
#define RT_SHADE_WPB (RT_SHADE_BLOCK / 64u)
#ifdef RT_SHADE_WAVES   /* waves per SIMD the register allocator must make room for (default: whatever 148 VGPRs allow = 3) */
#define RT_SHADE_OCCUPANCY __attribute__((amdgpu_waves_per_eu(RT_SHADE_WAVES, RT_SHADE_WAVES)))
#else
#define RT_SHADE_OCCUPANCY
#endif
#ifndef RT_SHADE_TILES
#define RT_SHADE_TILES 1u   /* measured: 1 tile 1 955, 4 tiles 1 908, 8 tiles 1 909 Mrays/s (per-wave atomics: 1 838) */
#endif
RT_DEV void push_queue(bool want, uint32_t path, uint32_t* __restrict__ queue, uint32_t* __restrict__ count) {
    const uint64_t m = __ballot(want);
    if (m == 0ull) return;
    uint32_t base = 0;
    const uint32_t lane = lane_id();
    const uint32_t leader = (uint32_t)__builtin_ctzll(m);
    if (lane == leader) base = atomicAdd(count, (uint32_t)__popcll(m));
    base = (uint32_t)__shfl((int)base, (int)leader, 64);
    if (want) queue[base + (uint32_t)__popcll(m & ((1ull << lane) - 1ull))] = path;
}

// ---- ray-queue compaction BY MATERIAL (scenes with more than one kind of scattering material) ---------------------------------
// k_shade's cost per wave is the sum of the material branches its 64 paths take (Lambertian: cosine pdf + light sample; Metal: a
// rejection loop; Dielectric: Fresnel; DiffuseLight and misses: terminal).  After a traversal launch the queue's slots are therefore
// regrouped by the class of what they hit -- 0 miss, 1 + rt_material.kind otherwise -- so that a wave shades one class:
// Three launches, no atomics, stable (slots of one class keep their queue order, so the sort's coherence survives and the order is
// a pure function of the queue): k_bin_count -- a wave takes `rounds` x 64 consecutive slots, writes each slot's class byte and its five
// ballot-popcount totals to unit_counts[class][unit]; k_bin_scan -- ONE workgroup turns that class-major table into its exclusive prefix
// (= class base + the unit's offset inside the class); k_bin_scatter -- the same waves replay their class bytes and place each slot at
// base[class] + prefix popcount of its ballot.  `order[i]` = queue slot shaded by thread i (hit records stay indexed by slot).
// (A first version with one device-scope atomic per wave and class was 5x SLOWER than not binning: same-address atomics from 8 XCDs
// serialise at about 12 ns each.)
#define RT_N_CLASSES 5u
RT_DEV uint32_t hit_class(const SceneDev& sc, const PathHit& ph) {
    if (ph.prim == RT_NONE) return 0u;
    return 1u + sc.materials[sc.prims[ph.prim].material].kind;
}
__global__ __launch_bounds__(256) void k_bin_count(SceneDev sc, const PathHit* __restrict__ hit, uint32_t n, uint32_t nunits, uint32_t rounds,
                                                   uint8_t* __restrict__ cls, uint32_t* __restrict__ unit_counts) {
    const uint32_t unit = (blockIdx.x * blockDim.x + threadIdx.x) >> 6, lane = lane_id();
    if (unit >= nunits) return;
    uint32_t cnt[RT_N_CLASSES] = {0u, 0u, 0u, 0u, 0u};
    for (uint32_t r = 0; r < rounds; ++r) {
        const uint32_t slot = (unit * rounds + r) * 64u + lane;
        uint32_t c = RT_N_CLASSES;
        if (slot < n) { c = hit_class(sc, hit[slot]); cls[slot] = (uint8_t)c; }
#pragma unroll
        for (uint32_t k = 0; k < RT_N_CLASSES; ++k) cnt[k] += (uint32_t)__popcll(__ballot(c == k));
    }
#pragma unroll
    for (uint32_t k = 0; k < RT_N_CLASSES; ++k) if (lane == k) unit_counts[(size_t)k * nunits + unit] = cnt[k];
}
// exclusive prefix of `total` counts in place, one workgroup of 1024 threads, 4 entries per thread and tile, carry across tiles
__global__ __launch_bounds__(1024) void k_bin_scan(uint32_t* __restrict__ v, uint32_t total) {
    __shared__ uint32_t wsum[16];
    __shared__ uint32_t carry_s;
    const uint32_t tid = threadIdx.x, lane = tid & 63u, w = tid >> 6;
    if (tid == 0) carry_s = 0u;
    __syncthreads();
    for (uint32_t t0 = 0; t0 < total; t0 += 4096u) {
        const uint32_t i = t0 + tid * 4u;
        uint32_t e[4];
#pragma unroll
        for (uint32_t k = 0; k < 4u; ++k) e[k] = i + k < total ? v[i + k] : 0u;
        const uint32_t mine = e[0] + e[1] + e[2] + e[3];
        uint32_t inc = mine;   // inclusive scan inside the wave
#pragma unroll
        for (uint32_t d = 1; d < 64u; d <<= 1) { const uint32_t o = (uint32_t)__shfl_up((int)inc, d, 64); if (lane >= d) inc += o; }
        if (lane == 63u) wsum[w] = inc;
        __syncthreads();
        uint32_t before = carry_s;
        for (uint32_t k = 0; k < w; ++k) before += wsum[k];
        uint32_t run = before + inc - mine;
#pragma unroll
        for (uint32_t k = 0; k < 4u; ++k) { if (i + k < total) v[i + k] = run; run += e[k]; }
        __syncthreads();
        if (tid == 1023u) carry_s = run;
        __syncthreads();
    }
}
__global__ __launch_bounds__(256) void k_bin_scatter(const uint8_t* __restrict__ cls, uint32_t n, uint32_t nunits, uint32_t rounds,
                                                     const uint32_t* __restrict__ unit_base, uint32_t* __restrict__ order) {
    const uint32_t unit = (blockIdx.x * blockDim.x + threadIdx.x) >> 6, lane = lane_id();
    if (unit >= nunits) return;
    uint32_t base[RT_N_CLASSES];
#pragma unroll
    for (uint32_t k = 0; k < RT_N_CLASSES; ++k) base[k] = unit_base[(size_t)k * nunits + unit];
    const uint64_t below = (1ull << lane) - 1ull;
    for (uint32_t r = 0; r < rounds; ++r) {
        const uint32_t slot = (unit * rounds + r) * 64u + lane;
        const uint32_t c = slot < n ? (uint32_t)cls[slot] : RT_N_CLASSES;
#pragma unroll
        for (uint32_t k = 0; k < RT_N_CLASSES; ++k) {
            const uint64_t m = __ballot(c == k);
            if (c == k) order[base[k] + (uint32_t)__popcll(m & below)] = slot;
            base[k] += (uint32_t)__popcll(m);
        }
    }
}

#define RT_SHADE_PARAMS SceneDev sc, PathBuffers pb, const uint32_t* __restrict__ queue, const uint32_t* __restrict__ order, const uint32_t* __restrict__ n_ptr, \
                        uint32_t bounce, int32_t max_depth, int32_t integrator, uint64_t seed,                                                                    \
                        uint32_t* __restrict__ next_queue, uint32_t* __restrict__ next_keys, uint32_t* __restrict__ next_count,                                    \
                        uint32_t* __restrict__ shadow_queue, uint32_t* __restrict__ shadow_keys, uint32_t* __restrict__ shadow_count
#define RT_SHADE_ARGS sc, pb, queue, order, n_ptr, bounce, max_depth, integrator, seed, next_queue, next_keys, next_count, shadow_queue, shadow_keys, shadow_count
RT_DEV void shade_block(const SceneDev& sc, const PathBuffers& pb, const uint32_t* __restrict__ queue, const uint32_t* __restrict__ order, const uint32_t* __restrict__ n_ptr,
                        uint32_t bounce, int32_t max_depth, int32_t integrator, uint64_t seed,
                        uint32_t* __restrict__ next_queue, uint32_t* __restrict__ next_keys, uint32_t* __restrict__ next_count,
                        uint32_t* __restrict__ shadow_queue, uint32_t* __restrict__ shadow_keys, uint32_t* __restrict__ shadow_count) {
    const uint32_t n = *n_ptr;
    // A block shades RT_SHADE_TILES tiles of 256 queue slots one after the other and parks what each thread wants to push in LDS; the
    // queues are then extended with ONE atomic per block and queue (push_tiles_block).  Whole blocks leave together: every wave of a
    // live block takes part in the barriers.
    const uint32_t block_first = blockIdx.x * (RT_SHADE_BLOCK * RT_SHADE_TILES);
    if (block_first >= n) return;
    __shared__ uint32_t st_path[RT_SHADE_TILES][RT_SHADE_BLOCK], st_nkey[RT_SHADE_TILES][RT_SHADE_BLOCK], st_skey[RT_SHADE_TILES][RT_SHADE_BLOCK];
    __shared__ uint8_t st_flag[RT_SHADE_TILES][RT_SHADE_BLOCK];
    const uint32_t cap = pb.cap;
#pragma unroll 1
    for (uint32_t tile = 0; tile < RT_SHADE_TILES; ++tile) {
    const uint32_t gid = block_first + tile * RT_SHADE_BLOCK + threadIdx.x;
    const bool valid = gid < n;
    bool want_next = false, want_shadow = false;
    uint32_t path = 0, next_key = 0, shadow_key = 0;
    if (valid) {
        const uint32_t slot = order ? order[gid] : gid;   // material-binned order (k_bin_scatter) or the queue's own
        path = queue[slot];
        Ray ray;
        const double4* r4 = (const double4*)(pb.rays + (size_t)path * RT_RAY_STRIDE);
        const double4 ra = r4[0], rb = r4[1];
        ray.o = mk(ra.x, ra.y, ra.z); ray.d = mk(ra.w, rb.x, rb.y); ray.time = rb.z;
        const PathHit ph = pb.hit[slot];   // written by k_trace per slot of this queue
        const uint32_t prim = ph.prim;
        d3 terminal = mk(0.0, 0.0, 0.0);
        bool done = false;
        HitRecord hr;
        unsigned long long sg = 0ull;   // this level's events (rt_math.h "path signatures"); only summed when pb.sig is set
        const bool want_sig = pb.sig != nullptr;
        if (want_sig) {
            unsigned long long who = 0xFFFFFFFFFFFFFFFFull;
            if (prim != RT_NONE) {
                const rt_prim hp = sc.prims[prim];
                const uint32_t inst_id = ph.inst != RT_NONE ? sc.prims[ph.inst].inst_id : hp.inst_id;
                who = (unsigned long long)hp.object_id | ((unsigned long long)inst_id << 32);
            }
            sg += sig_hash(bounce, SIG_HIT, who);
        }
        if (prim == RT_NONE) { terminal = background_colour(sc.background, ray); done = true; }  // integrator.rs:39-41
        else if (!full_hit(sc, prim, ph.inst, ray, 0.001, ph.t, hr)) { terminal = mk(__builtin_nan(""), __builtin_nan(""), __builtin_nan("")); done = true; }  // cannot happen: poison loudly
        if (!done) {
            const rt_material mat = sc.materials[hr.material];
            Pdf material_pdf;
            if (want_sig) {
                sg += sig_hash(bounce, SIG_FRONT, hr.front_face ? 1ull : 0ull);
                if (mat.kind == RT_MAT_LAMBERTIAN || mat.kind == RT_MAT_DIFFUSE_LIGHT) sg += sig_hash(bounce, SIG_TEX, tex_sig(sc, mat.texture, hr.u, hr.v, hr.point));
            }
            if (!scattering_pdf(mat, ray, hr, material_pdf)) {  // DiffuseLight: return emitted (integrator.rs:45-50)
                terminal = material_emitted(sc, hr.material, hr.u, hr.v, hr.point);
                done = true;
                if (want_sig) sg += sig_hash(bounce, SIG_KIND, 0ull);
            } else {
                const PathInfo pi = pb.info[path];
                Rng rng(seed, pi.pixel, pi.sample, bounce + 1u);
                const double nan = __builtin_nan("");
                d3 brdf, out_dir, c_l = mk(0.0, 0.0, 0.0);
                double cos_theta, pdf = 1.0, w = 0.0, t_light = nan;
                const bool delta = pdf_is_delta(material_pdf);
                if (want_sig) sg += sig_hash(bounce, SIG_KIND, delta ? 1ull : 2ull);
                d3 gen_dir = mk(0.0, 0.0, 0.0);   // what the scatter pdf generated, before any normalisation (signature: reflect or refract)
                if (integrator == RT_INT_MIS) {
                    if (delta) {  // integrator.rs:54-66
                        gen_dir = pdf_generate(material_pdf, rng);
                        out_dir = normalize(gen_dir);
                        cos_theta = dot(out_dir, hr.normal);
                        brdf = material_brdf(sc, mat, hr, out_dir);
                    } else {
                        bool have_light = false;
                        double light_pdf_value = 0.0;
                        uint32_t li = 0;
                        d3 light_dir = mk(0, 0, 0);
                        if (sc.n_lights > 0u) {  // integrator.rs:68-88
                            li = rng.index(sc.n_lights);
                            const rt_light lt = sc.lights[li];
                            Pdf light_pdf = light_pdf_for_point(lt, hr.point);
                            light_dir = normalize(pdf_generate(light_pdf, rng));
                            if (!(dot(hr.normal, light_dir) < 0.0)) {
                                double tl;
                                if (sphere_hit(ld3(lt.center), lt.radius, hr.point, light_dir, 0.001, 10000.0, tl)) {
                                    have_light = true; t_light = tl;
                                    light_pdf_value = pdf_value(light_pdf, light_dir);
                                }
                            }
                            if (want_sig) sg += sig_hash(bounce, SIG_LIGHT, 2ull * li + (have_light ? 1ull : 0ull));
                        }
                        out_dir = normalize(pdf_generate(material_pdf, rng));  // integrator.rs:95 / 158
                        cos_theta = dot(out_dir, hr.normal);
                        pdf = pdf_value(material_pdf, out_dir);
                        brdf = material_brdf(sc, mat, hr, out_dir);
                        if (have_light) {  // the visible-branch terms (integrator.rs:99-152), used by k_unwind only if the shadow test passes
                            const rt_light lt = sc.lights[li];
                            d3 lp = hr.point + light_dir * t_light;          // light_hit.point is unused; (u,v) feed emitted()
                            d3 ln = (lp - ld3(lt.center)) / lt.radius;
                            double lu = 0.0, lv = 0.0;
                            if (material_uses_uv(sc, lt.material)) spherical_uv(ln, lu, lv);
                            d3 light_emit = material_emitted(sc, lt.material, lu, lv, hr.point);
                            double th;
                            bool material_ray_hit_light = sphere_hit(ld3(lt.center), lt.radius, hr.point, out_dir, 0.001, 10000.0, th);
                            w = power_heuristic(pdf, material_ray_hit_light ? light_pdf_value : 0.0);
                            double light_weight = power_heuristic(light_pdf_value, pdf_value(material_pdf, light_dir));
                            double light_cos_theta = fmax(dot(hr.normal, light_dir), 0.0);
                            c_l = material_brdf(sc, mat, hr, light_dir) * light_cos_theta * light_emit * light_weight / light_pdf_value;
                            double4* s4 = (double4*)(pb.shadow_rays + (size_t)path * RT_RAY_STRIDE);
                            s4[0] = make_double4(hr.point.x, hr.point.y, hr.point.z, light_dir.x); s4[1] = make_double4(light_dir.y, light_dir.z, ray.time, t_light);  // last slot: t_light for the visibility walk's early exit
                            want_shadow = true;
                            shadow_key = sort_key(sc, hr.point, light_dir);
                        }
                    }
                } else if (integrator == RT_INT_LIGHT_IS) {  // integrator.rs:187-224 (emitted == 0 here)
                    Pdf scatter_pdf = material_pdf;
                    if (!delta && sc.n_lights > 0u) {
                        const uint32_t li = rng.index(sc.n_lights);
                        scatter_pdf = light_pdf_for_point(sc.lights[li], hr.point);
                        if (want_sig) sg += sig_hash(bounce, SIG_LIGHT, 2ull * li);
                    }
                    out_dir = pdf_generate(scatter_pdf, rng);
                    gen_dir = out_dir;
                    cos_theta = dot(out_dir, hr.normal);
                    if (cos_theta < 0.0) { terminal = mk(0.0, 0.0, 0.0); done = true; }
                    if (want_sig) sg += sig_hash(bounce, SIG_CUT, cos_theta < 0.0 ? 1ull : 0ull);
                    pdf = pdf_is_delta(scatter_pdf) ? 1.0 : pdf_value(scatter_pdf, out_dir);
                    brdf = material_brdf(sc, mat, hr, out_dir);
                } else {  // integrator.rs:273-291 (BRDF) / 340-361 (uniform)
                    Pdf scatter_pdf = material_pdf;
                    if (integrator == RT_INT_UNIFORM && !delta) { scatter_pdf.kind = PDF_UNIFORM_HEMI; scatter_pdf.a = hr.normal; }
                    out_dir = pdf_generate(scatter_pdf, rng);
                    gen_dir = out_dir;
                    cos_theta = dot(out_dir, hr.normal);
                    pdf = pdf_is_delta(scatter_pdf) ? 1.0 : pdf_value(scatter_pdf, out_dir);
                    brdf = material_brdf(sc, mat, hr, out_dir);
                }
                if (want_sig && material_pdf.kind == PDF_DIELECTRIC)   // reflected (1) or refracted (0): pdf.rs:167-173
                    sg += sig_hash(bounce, SIG_DELTA, (gen_dir.x == material_pdf.a.x && gen_dir.y == material_pdf.a.y && gen_dir.z == material_pdf.a.z) ? 1ull : 0ull);
                if (!done) {
                    double* rec = pb.records + ((size_t)bounce * cap + path) * REC_STRIDE;
                    rec[REC_BRDF_X] = brdf.x; rec[REC_BRDF_Y] = brdf.y; rec[REC_BRDF_Z] = brdf.z;
                    rec[REC_COS] = cos_theta; rec[REC_PDF] = pdf; rec[REC_W] = w;
                    rec[REC_CL_X] = c_l.x; rec[REC_CL_Y] = c_l.y; rec[REC_CL_Z] = c_l.z;
                    rec[REC_T_LIGHT] = t_light;
                    pb.info[path].nb = bounce + 1u;
                    if ((int32_t)(bounce + 1u) >= max_depth) {  // next level has depth <= 0: returns 0 without tracing (integrator.rs:33-35)
                        terminal = mk(0.0, 0.0, 0.0);
                        done = true;
                    } else {
                        double4* n4 = (double4*)(pb.rays + (size_t)path * RT_RAY_STRIDE);
                        n4[0] = make_double4(hr.point.x, hr.point.y, hr.point.z, out_dir.x); n4[1] = make_double4(out_dir.y, out_dir.z, ray.time, 0.0);
                        want_next = true;
                        next_key = sort_key(sc, hr.point, out_dir);
                    }
                }
            }
        }
        if (want_sig) pb.sig[path] += sg;
        if (done) { pb.terminal[(size_t)path * 4u + 0] = terminal.x; pb.terminal[(size_t)path * 4u + 1] = terminal.y; pb.terminal[(size_t)path * 4u + 2] = terminal.z; }
    }
    st_path[tile][threadIdx.x] = path; st_nkey[tile][threadIdx.x] = next_key; st_skey[tile][threadIdx.x] = shadow_key;
    st_flag[tile][threadIdx.x] = (uint8_t)((want_next ? 1u : 0u) | (want_shadow ? 2u : 0u));
    }
    // every thread reads back only what it parked itself: no barrier needed before this point
    __shared__ uint32_t cnt_s[2][RT_SHADE_TILES * RT_SHADE_WPB], base_s[2];
    const uint32_t lane = lane_id(), wv = threadIdx.x >> 6;
#pragma unroll
    for (uint32_t tile = 0; tile < RT_SHADE_TILES; ++tile) {
        const uint32_t fl = st_flag[tile][threadIdx.x];
        const uint64_t ma = __ballot((fl & 1u) != 0u), mb = __ballot((fl & 2u) != 0u);
        if (lane == 0u) { cnt_s[0][tile * RT_SHADE_WPB + wv] = (uint32_t)__popcll(ma); cnt_s[1][tile * RT_SHADE_WPB + wv] = (uint32_t)__popcll(mb); }
    }
    __syncthreads();
    if (threadIdx.x < 2u) {
        uint32_t tot = 0u;
        for (uint32_t k = 0; k < RT_SHADE_TILES * RT_SHADE_WPB; ++k) tot += cnt_s[threadIdx.x][k];
        base_s[threadIdx.x] = tot ? atomicAdd(threadIdx.x == 0u ? next_count : shadow_count, tot) : 0u;
    }
    __syncthreads();
    uint32_t ra = base_s[0], rb = base_s[1];   // running bases over (tile, wave) in that order
#pragma unroll
    for (uint32_t tile = 0; tile < RT_SHADE_TILES; ++tile) {   // slots inside a block: tile, then wave, then lane
        uint32_t ba = 0u, bb = 0u;
#pragma unroll
        for (uint32_t k = 0; k < RT_SHADE_WPB; ++k) {
            if (k == wv) { ba = ra; bb = rb; }
            ra += cnt_s[0][tile * RT_SHADE_WPB + k]; rb += cnt_s[1][tile * RT_SHADE_WPB + k];
        }
        const uint32_t fl = st_flag[tile][threadIdx.x], path = st_path[tile][threadIdx.x];
        const uint64_t ma = __ballot((fl & 1u) != 0u), mb = __ballot((fl & 2u) != 0u);
        if (fl & 1u) { const uint32_t sl = ba + (uint32_t)__popcll(ma & ((1ull << lane) - 1ull)); next_queue[sl] = path; if (next_keys) next_keys[sl] = st_nkey[tile][threadIdx.x]; }
        if (fl & 2u) { const uint32_t sl = bb + (uint32_t)__popcll(mb & ((1ull << lane) - 1ull)); shadow_queue[sl] = path; if (shadow_keys) shadow_keys[sl] = st_skey[tile][threadIdx.x]; }
    }
}
// Two builds of the same body.  k_shade: the allocator's own choice (148 VGPRs, 3 waves per SIMD) -- for the scenes whose frame is bound by k_trace's
// issue slots, where a faster k_shade only hands its time to the k_trace of the other lane.  k_shade_dense: 4 waves per SIMD (128 VGPRs + 88 B of
// scratch), a third faster on its own -- for the small scenes (no ray sorting: earth, broken cube), where shading is the larger half of the frame:
// +4 % there, -2 % on the pagoda.
__global__ __launch_bounds__(RT_SHADE_BLOCK) RT_SHADE_OCCUPANCY void k_shade(RT_SHADE_PARAMS) { shade_block(RT_SHADE_ARGS); }
__global__ __launch_bounds__(RT_SHADE_BLOCK) __attribute__((amdgpu_waves_per_eu(4, 4))) void k_shade_dense(RT_SHADE_PARAMS) { shade_block(RT_SHADE_ARGS); }

// ------------------------------------------------------------------------------------------------
// k_unwind / k_accumulate / k_resolve
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_unwind(PathBuffers pb, uint32_t n) {
    const uint32_t p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= n) return;
    const size_t cap = pb.cap;
    d3 L = mk(pb.terminal[(size_t)p * 4u + 0], pb.terminal[(size_t)p * 4u + 1], pb.terminal[(size_t)p * 4u + 2]);
    unsigned long long sg = 0ull;
    for (int k = (int)pb.info[p].nb - 1; k >= 0; --k) {
        const double* rec = pb.records + ((size_t)k * cap + p) * REC_STRIDE;
        d3 brdf = mk(rec[REC_BRDF_X], rec[REC_BRDF_Y], rec[REC_BRDF_Z]);
        double cos_theta = rec[REC_COS], pdf = rec[REC_PDF], t_light = rec[REC_T_LIGHT];
        bool visible = false;
        if (t_light == t_light) {  // a light sample was taken: visible iff the shadow ray hit and |t_scene - t_light| <= 1e-4 (integrator.rs:82-86)
            double t_shadow = rec[REC_T_SHADOW];
            visible = (t_shadow == t_shadow) && !(fabs(t_shadow - t_light) > 0.0001);
            sg += sig_hash((uint32_t)k, SIG_VIS, visible ? 1ull : 0ull);
        }
        d3 e = mk(0.0, 0.0, 0.0);  // emitted of a scattering material
        if (visible) {
            bool has_light = L.x >= 0.001 && L.y >= 0.001 && L.z >= 0.001;  // integrator.rs:119-121
            sg += sig_hash((uint32_t)k, SIG_GATE, has_light ? 1ull : 0ull);
            double w = has_light ? rec[REC_W] : 0.0;
            d3 material_contribution = brdf * cos_theta * L * w / pdf;
            d3 c_l = mk(rec[REC_CL_X], rec[REC_CL_Y], rec[REC_CL_Z]);
            L = e + material_contribution + c_l;
        } else {
            L = e + (brdf * cos_theta * L) / pdf;
        }
    }
    pb.radiance[0 * cap + p] = L.x; pb.radiance[1 * cap + p] = L.y; pb.radiance[2 * cap + p] = L.z;
    if (pb.sig) pb.sig[p] += sg;
}

// per pixel: fixed-order f64 sum over this pass's samples, then one fp32 add into the accumulator
__global__ __launch_bounds__(256) void k_accumulate(const double* __restrict__ radiance, const PathInfo* __restrict__ info, uint32_t cap, uint32_t npix, uint32_t samples,
                                                    float* __restrict__ accum) {
    const uint32_t q = blockIdx.x * blockDim.x + threadIdx.x;  // slot of the sample layer (k_raygen: every layer uses the same slot -> pixel map)
    if (q >= npix) return;
    const uint32_t px = info[q].pixel;
    double r = 0.0, g = 0.0, b = 0.0;
    for (uint32_t s = 0; s < samples; ++s) {
        size_t p = (size_t)s * npix + q;
        r += radiance[0 * (size_t)cap + p]; g += radiance[1 * (size_t)cap + p]; b += radiance[2 * (size_t)cap + p];
    }
    accum[3 * (size_t)px + 0] += (float)r; accum[3 * (size_t)px + 1] += (float)g; accum[3 * (size_t)px + 2] += (float)b;
}

RT_DEV float aces_channel(float x) {  // main.rs:913-921
    float v = (x * (2.51f * x + 0.03f)) / (x * (2.43f * x + 0.59f) + 0.14f);
    return fminf(fmaxf(v, 0.0f), 1.0f);  // glam clamp = max(min).min(max): NaN -> 0
}
__global__ __launch_bounds__(256) void k_resolve(const float* __restrict__ accum, uint32_t n3, float inv_unused, uint32_t spp, int tonemap, float* __restrict__ out) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n3) return;
    float c = accum[i] / (float)spp;  // main.rs:1009
    if (tonemap) c = aces_channel(c);
    if (c != c) c = 0.0f;             // image.rs:48-56
    out[i] = c;
}

__global__ void k_aos_to_soa(const double* __restrict__ aos, uint32_t n, double* __restrict__ soa, uint32_t cap) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
#pragma unroll
    for (int c = 0; c < 7; ++c) soa[(size_t)i * RT_RAY_STRIDE + c] = aos[(size_t)i * 7 + c];
}
__global__ void k_soa_to_aos(const double* __restrict__ soa, uint32_t n, uint32_t cap, double* __restrict__ aos) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
#pragma unroll
    for (int c = 0; c < 7; ++c) aos[(size_t)i * 7 + c] = soa[(size_t)i * RT_RAY_STRIDE + c];
}
__global__ void k_iota(uint32_t* q, uint32_t n) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) q[i] = i;
}
__global__ void k_set_u32(uint32_t* p, uint32_t v) { *p = v; }
// rt_pdf_eval: PDF::value / PDF::generate of pdf.rs for n independent items, each a list of n_comp PDFs (n_comp > 1: MixturePDF Uniform)
#define RT_PDF_MAX_COMP 8u
__global__ __launch_bounds__(256) void k_pdf_eval(uint32_t n, uint32_t n_comp, const rt_pdf_desc* __restrict__ descs, const double* __restrict__ dirs, uint64_t seed,
                                                 double* __restrict__ value, double* __restrict__ generated) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    Pdf pdfs[RT_PDF_MAX_COMP];
    for (uint32_t c = 0; c < n_comp; ++c) {
        const rt_pdf_desc d = descs[(size_t)i * n_comp + c];
        pdfs[c].kind = (int)d.kind; pdfs[c].a = ld3(d.a); pdfs[c].b = ld3(d.b); pdfs[c].s = d.s; pdfs[c].cannot_refract = d.cannot_refract != 0u;
    }
    const d3 dir = ld3(dirs + 3 * (size_t)i);
    Rng rng(seed, i, 0u, 0u);
    const double v = n_comp == 1u ? pdf_value(pdfs[0], dir) : pdf_mixture_value(pdfs, n_comp, dir);
    const d3 g = n_comp == 1u ? pdf_generate(pdfs[0], rng) : pdf_mixture_generate(pdfs, n_comp, rng);
    value[i] = v;
    generated[3 * (size_t)i] = g.x; generated[3 * (size_t)i + 1] = g.y; generated[3 * (size_t)i + 2] = g.z;
}
__global__ void k_div_sqrt(uint32_t n, const double* a, const double* b, double* q, double* s) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) { q[i] = a[i] / b[i]; s[i] = __builtin_sqrt(a[i]); }
}

// ------------------------------------------------------------------------------------------------
// host side of the ABI
// ------------------------------------------------------------------------------------------------
// Scratch of rt_trace_closest_device_noalloc (the entry point for callers that cannot allocate): the statistics
// counters k_trace adds to, then one 64-byte ray block per ray (the layout k_trace reads).  Every GPU the library runs
// on holds its own copy of this array until the process ends.  In host code the NAME is only a stand-in: the device
// address is taken once per device with hipGetSymbolAddress (rt_context_create) and kept in rt_context::scratch.
#define RT_NOALLOC_MAX_RAYS (1u << 18)
#define RT_NOALLOC_STATS_BYTES 256u
static __device__ __align__(256) unsigned char scratch[RT_NOALLOC_STATS_BYTES + (size_t)RT_NOALLOC_MAX_RAYS * RT_RAY_STRIDE * sizeof(double)];

struct rt_context {
    int device;
    hipStream_t stream;
    bool own_stream;
    unsigned char* scratch;  // this device's copy of `scratch` (device address)
    int live_scenes = 0;     // rt_scene handles that still point at this context (rt_context_destroy refuses while > 0)
};

struct Workspace {
    PathBuffers pb{};
    uint32_t* queue[2] = {nullptr, nullptr};
    uint32_t* shadow_queue = nullptr;
    uint32_t* counts = nullptr;  // [0],[1]: path queue counts; [2]: shadow count
    unsigned long long* stats = nullptr;  // rays, box, tri, sph
    uint32_t* in_pixel = nullptr; uint32_t* in_sample = nullptr;
    uint32_t* raw_queue = nullptr; uint32_t* raw_keys = nullptr; uint32_t* raw_squeue = nullptr; uint32_t* raw_skeys = nullptr; uint32_t* sorted_keys = nullptr;  // ray sorting
    void* sort_tmp = nullptr; size_t sort_tmp_bytes = 0;
    uint32_t cap = 0; int32_t depth = 0;
    uint32_t* order = nullptr;      // material-binned shading order of the current queue (k_bin_scatter)
    uint8_t* cls = nullptr;         // class byte per queue slot (k_bin_count)
    uint32_t* bins = nullptr;       // [class][unit] counts, then their exclusive prefix (k_bin_count / k_bin_scan)
    uint4* walk_state = nullptr;    // k_walk: [RT_WALK_STATE_VEC][cap] per-ray filter state (k_walk_setup)
    uint32_t* walk_odd = nullptr;   // queue slots k_walk leaves to k_trace (rays that need the literal box test)
    uint32_t* walk_ctr = nullptr;   // [0] next queue slot to hand out, [1] entries of walk_odd
    uint32_t* h_counts = nullptr;  // pinned host copy of counts (the live ray count of the next bounce sizes its sort and launches)
    std::vector<void*> allocs;
};

struct rt_scene {
    rt_context* ctx;
    void* dblob = nullptr;
    uint8_t* d_mat_uses_uv = nullptr;
    uint8_t* d_stream = nullptr;
    uint32_t* d_node_off = nullptr;
    uint8_t* d_stream_full = nullptr;
    uint32_t* d_node_off_full = nullptr;
    uint32_t* d_cert_cells = nullptr;
    CertEntry* d_cert_entries = nullptr;
    double* d_cert_normals = nullptr;
    double* d_cert_planes = nullptr;
    uint32_t* d_cert_plane_off = nullptr;
    uint32_t* d_cert_plane_leaf = nullptr;
    SceneDev dev{};
    rt_blob_header hdr{};
    std::vector<rt_prim> host_prims;
    bool has_instances = false;
    bool reg_ray = false;   // most leaves are not triangles: k_trace variant with the f64 ray in registers
    bool sortable = false;  // scene has a finite box to quantise ray origins against
    bool walk_ok = false;   // k_walk (per-lane walk) may trace this scene's secondary and shadow rays
    bool bin_materials = false;  // more than one kind of scattering material: shade in material-binned order
    int n_cu = 256;
    Workspace ws;
    Workspace ws2;                  // second render lane (rt_render runs two passes at a time, each on its own stream)
    hipStream_t stream2 = nullptr;
    uint64_t ws2_failed_cap = ~0ull;  // smallest second-lane size (paths) that could not be allocated
    // rt_render_begin / rt_render_end: at most one job per lane in flight; jobs add to the accumulator in the order they were begun
    struct rt_job* lane_job[2] = {nullptr, nullptr};
    uint64_t job_seq = 0;               // jobs begun so far
    hipEvent_t job_done[4] = {nullptr, nullptr, nullptr, nullptr};   // job k records job_done[k % 4] after its last accumulate
    std::mutex job_mu; std::condition_variable job_cv; uint64_t jobs_recorded = 0;   // jobs whose done-event has been recorded (host side)
    std::thread worker[2];              // one host thread per lane, started by the first rt_render_begin on that lane (a fresh thread per job costs HIP's per-thread set-up every time)
    bool workers_quit = false;
};

struct rt_job {
    rt_scene* sc = nullptr; rt_render_params p{}; float* accum = nullptr; int lane = 0; uint64_t seq = 0; uint32_t spass = 1;
    bool started = false, finished = false;   // under rt_scene::job_mu
    int rc = RT_OK; std::string err; rt_stats st{};
};
static bool scene_busy(rt_scene* sc) { return sc->lane_job[0] != nullptr || sc->lane_job[1] != nullptr; }
#define RT_NOT_BUSY(sc, who) do { if (scene_busy(sc)) return fail(RT_ERR_BAD_ARG, who ": render jobs begun with rt_render_begin are still in flight on this scene (rt_render_end them first)"); } while (0)

static void ws_free(Workspace& ws) {
    for (void* p : ws.allocs) (void)hipFree(p);
    if (ws.h_counts) (void)hipHostFree(ws.h_counts);
    ws = Workspace();
}
template <class T> static int ws_alloc(Workspace& ws, T*& out, size_t count) {
    void* p = nullptr;
    HIP_OK(hipMalloc(&p, std::max<size_t>(count, 1) * sizeof(T)));
    ws.allocs.push_back(p);
    out = (T*)p;
    return RT_OK;
}
static int ws_ensure(rt_scene* sc, Workspace& ws, uint32_t cap, int32_t depth);
static int ws_ensure(rt_scene* sc, uint32_t cap, int32_t depth) { return ws_ensure(sc, sc->ws, cap, depth); }
static int ws_ensure(rt_scene* sc, Workspace& ws, uint32_t cap, int32_t depth) {
    if (ws.cap >= cap && ws.depth >= depth) return RT_OK;
    ws_free(ws);
    cap = (cap + 255u) & ~255u;
    int rc;
#define WS(ptr, count) if ((rc = ws_alloc(ws, ptr, (size_t)(count))) != RT_OK) { ws_free(ws); return rc; }
    WS(ws.pb.rays, RT_RAY_STRIDE * (size_t)cap) WS(ws.pb.shadow_rays, RT_RAY_STRIDE * (size_t)cap)
    WS(ws.pb.hit_prim, cap) WS(ws.pb.hit_inst, cap) WS(ws.pb.hit_t, cap)
    WS(ws.pb.hit, cap) WS(ws.pb.info, cap)
    WS(ws.pb.terminal, 4 * (size_t)cap) WS(ws.pb.radiance, 3 * (size_t)cap)
    WS(ws.pb.records, (size_t)std::max(depth, 1) * REC_STRIDE * cap)
    WS(ws.queue[0], cap) WS(ws.queue[1], cap) WS(ws.shadow_queue, cap) WS(ws.counts, 4) WS(ws.stats, 16)
    WS(ws.in_pixel, cap) WS(ws.in_sample, cap)
    WS(ws.raw_queue, cap) WS(ws.raw_keys, cap) WS(ws.raw_squeue, cap) WS(ws.raw_skeys, cap) WS(ws.sorted_keys, cap)
    {
        size_t bytes = 0;
        hipError_t e = rocprim::radix_sort_pairs(nullptr, bytes, ws.raw_keys, ws.sorted_keys, ws.raw_queue, ws.queue[0], (size_t)cap, 0u, RT_SORT_BITS, sc->ctx->stream);
        if (e != hipSuccess) { ws_free(ws); return fail(RT_ERR_HIP, std::string("rocprim size query: ") + hipGetErrorString(e)); }
        uint8_t* tmp = nullptr;
        WS(tmp, bytes)
        ws.sort_tmp = tmp; ws.sort_tmp_bytes = bytes;
    }
#undef WS
    if (hipHostMalloc((void**)&ws.h_counts, 4 * sizeof(uint32_t), hipHostMallocDefault) != hipSuccess) { ws_free(ws); return fail(RT_ERR_OOM, "pinned host allocation failed"); }
    ws.pb.cap = cap; ws.cap = cap; ws.depth = depth;
    return RT_OK;
}

static inline dim3 grid_for(uint32_t n) { return dim3((n + 255u) / 256u); }

struct EventTimer {
    struct Span { hipEvent_t a, b; int cat; };
    std::vector<Span> spans;
    hipStream_t s;
    bool enabled;
    explicit EventTimer(hipStream_t st, bool en) : s(st), enabled(en) {}
    void begin(int cat) {
        if (!enabled) return;
        Span sp; sp.cat = cat;
        (void)hipEventCreate(&sp.a); (void)hipEventCreate(&sp.b);
        (void)hipEventRecord(sp.a, s);
        spans.push_back(sp);
    }
    void end() { if (enabled) (void)hipEventRecord(spans.back().b, s); }
    void collect(double* ms_by_cat, int ncat, uint64_t* count_by_cat) {
        for (auto& sp : spans) {
            float ms = 0.f;
            (void)hipEventElapsedTime(&ms, sp.a, sp.b);
            if (sp.cat < ncat) { ms_by_cat[sp.cat] += ms; count_by_cat[sp.cat]++; }
            (void)hipEventDestroy(sp.a); (void)hipEventDestroy(sp.b);
        }
        spans.clear();
    }
};
enum { CAT_TRACE = 0, CAT_SHADE = 1, CAT_OTHER = 2, CAT_SHADOW = 3, CAT_N = 4 };  // CAT_SHADOW: traversal launches of MIS shadow rays (integrator.rs:82)

static void launch_trace(rt_scene* sc, unsigned long long* lstats, uint32_t lcap, hipStream_t st, bool count, const uint32_t* queue, const uint32_t* n_ptr, uint32_t n_bound, const double* rays, double t_min,
                         double t_max, uint32_t* out_prim, uint32_t* out_inst, double* out_t, uint32_t lanes = 64u, uint32_t t_stride = 1u, PathHit* out_hit = nullptr, bool shadow = false,
                         const uint32_t* remap = nullptr, bool count_steps = true) {
    if (n_bound == 0) return;
    SceneDev dv = sc->dev;   // this launch's copy of the kernel argument
    dv.count_steps = count_steps ? 1u : 0u;
    const uint32_t wpb = RT_TRACE_BLOCK / 64u;
    const dim3 grid((uint32_t)((((uint64_t)n_bound + lanes - 1u) / lanes + wpb - 1u) / wpb));  // waves of `lanes` rays, RT_TRACE_BLOCK / 64 waves per block
#define RT_LAUNCH_TRACE_S(C, I, R, S) hipLaunchKernelGGL((k_trace<C, I, R, S>), grid, dim3(RT_TRACE_BLOCK), 0, st, dv, queue, n_ptr, n_bound, rays, lcap, t_min, t_max, out_prim, out_inst, out_t, lstats, lanes, t_stride, out_hit, remap)
#define RT_LAUNCH_TRACE_R(C, I, R) do { if (shadow && !C) RT_LAUNCH_TRACE_S(C, I, R, !C); else RT_LAUNCH_TRACE_S(C, I, R, false); } while (0)
#define RT_LAUNCH_TRACE(C, I) do { if (sc->reg_ray) RT_LAUNCH_TRACE_R(C, I, true); else RT_LAUNCH_TRACE_R(C, I, false); } while (0)
    const bool inst = sc->has_instances;  // scenes without BVHNode objects use the kernel variant compiled without the nested walk
#ifdef RT_TRACE_DIAG
    unsigned long long d0[16];
    (void)hipMemcpyAsync(d0, lstats, sizeof d0, hipMemcpyDeviceToHost, st);
    (void)hipStreamSynchronize(st);
#endif
    if (count) { if (inst) RT_LAUNCH_TRACE(true, true); else RT_LAUNCH_TRACE(true, false); }
    else { if (inst) RT_LAUNCH_TRACE(false, true); else RT_LAUNCH_TRACE(false, false); }
#undef RT_LAUNCH_TRACE_S
#undef RT_LAUNCH_TRACE_R
#undef RT_LAUNCH_TRACE
#ifdef RT_TRACE_DIAG
    {
        unsigned long long d[16];
        (void)hipMemcpyAsync(d, lstats, sizeof d, hipMemcpyDeviceToHost, st);
        (void)hipStreamSynchronize(st);
        for (int k = 0; k < 16; ++k) if (k != 10) d[k] -= d0[k];
        (void)hipMemsetAsync(lstats + 10, 0, sizeof(unsigned long long), st);
        const double w = (double)std::max(d[8], 1ull);
        fprintf(stderr, "[diag] rays %llu waves %llu  steps/wave %.0f  leaf steps/wave %.0f  lane use %.1f%%  cycles/wave %.0f  cycles/step %.0f  wait/step %.0f  max steps %llu  exact box steps/wave %.1f  exact tri steps/wave %.1f  | of a wave's cycles: set-up %.1f%%  leaf steps %.1f%% (exact tri blocks %.1f%%)  exact box blocks %.1f%%  scalar-load wait %.1f%%\n",
                d[0], d[8], d[4] / w, d[5] / w, 100.0 * d[9] / (64.0 * std::max(d[4], 1ull)), d[7] / w, (double)d[7] / std::max(d[4], 1ull), (double)d[6] / std::max(d[4], 1ull), d[10], d[11] / w, d[12] / w,
                100.0 * d[1] / std::max(d[7], 1ull), 100.0 * d[2] / std::max(d[7], 1ull), 100.0 * d[13] / std::max(d[7], 1ull), 100.0 * d[14] / std::max(d[7], 1ull), 100.0 * d[6] / std::max(d[7], 1ull));
    }
#endif
}

static void launch_trace(rt_scene* sc, bool count, const uint32_t* queue, const uint32_t* n_ptr, uint32_t n_bound, const double* rays, double t_min,
                         double t_max, uint32_t* out_prim, uint32_t* out_inst, double* out_t) {
    launch_trace(sc, sc->ws.stats, sc->ws.cap, sc->ctx->stream, count, queue, n_ptr, n_bound, rays, t_min, t_max, out_prim, out_inst, out_t);
}

// Secondary / shadow rays of scenes k_walk serves: per-ray set-up, the per-lane walk on a persistent grid, and a k_trace launch
// over the (normally zero) rays that need the literal box test.
static int launch_walk(rt_scene* sc, Workspace& ws, hipStream_t st, const uint32_t* queue, const uint32_t* n_ptr, uint32_t n_bound, const double* rays, double t_min,
                       double t_max, PathHit* out_hit, double* out_t, uint32_t t_stride, bool shadow) {
    if (n_bound == 0) return RT_OK;
    if (!ws.walk_state) {   // first use on this workspace: 160 B of state per path + the left-over list (freed with the workspace)
        int rc;
        if ((rc = ws_alloc(ws, ws.walk_state, (size_t)RT_WALK_STATE_VEC * ws.cap)) != RT_OK || (rc = ws_alloc(ws, ws.walk_odd, ws.cap)) != RT_OK ||
            (rc = ws_alloc(ws, ws.walk_ctr, 4)) != RT_OK) { ws.walk_state = nullptr; return rc; }
    }
    HIP_OK(hipMemsetAsync(ws.walk_ctr, 0, 4 * sizeof(uint32_t), st));
    hipLaunchKernelGGL(k_walk_setup, grid_for(n_bound), dim3(256), 0, st, sc->dev, queue, n_ptr, rays, ws.walk_state, ws.cap, t_min, t_max, ws.walk_odd, ws.walk_ctr + 1, ws.stats);
    const uint32_t blocks = std::min<uint32_t>((n_bound + 255u) / 256u, (uint32_t)sc->n_cu * 4u);   // persistent: 4 blocks (16 waves) per CU
    if (shadow) hipLaunchKernelGGL((k_walk<true>), dim3(blocks), dim3(256), 0, st, sc->dev, queue, n_ptr, rays, ws.walk_state, ws.cap, t_min, t_max, out_hit, out_t, t_stride, ws.walk_ctr, ws.stats);
    else hipLaunchKernelGGL((k_walk<false>), dim3(blocks), dim3(256), 0, st, sc->dev, queue, n_ptr, rays, ws.walk_state, ws.cap, t_min, t_max, out_hit, out_t, t_stride, ws.walk_ctr, ws.stats);
    launch_trace(sc, ws.stats, ws.cap, st, false, queue, ws.walk_ctr + 1, n_bound, rays, t_min, t_max, nullptr, nullptr, out_t, 64u, t_stride, out_hit, shadow, ws.walk_odd);
    return RT_OK;
}

// Runs the wavefront loop for n paths whose (pixel, sample) come either from explicit device lists
// (ws.in_pixel / ws.in_sample) or from the frame formula; leaves per-path radiance in ws.pb.radiance.
// stable LSD radix sort of the n live (key, path) pairs of a queue
static int sort_queue(Workspace& ws, hipStream_t st, uint32_t n, uint32_t* keys_in, uint32_t* queue_in, uint32_t* out_queue) {
    size_t bytes = ws.sort_tmp_bytes;
    HIP_OK(rocprim::radix_sort_pairs(ws.sort_tmp, bytes, keys_in, ws.sorted_keys, queue_in, out_queue, (size_t)n, 0u, RT_SORT_BITS, st));
    return RT_OK;
}

// rt_capture_rays: copies out every stride-th ray of one queue of the render loop, with what the loop's own traversal launch reported for it
struct Capture {
    uint32_t bounce = 0; bool shadow = false; uint32_t stride = 1, max_n = 0;
    std::vector<double> rays, t, t_light; std::vector<uint32_t> prim, inst;
    uint32_t queue_len = 0;
};
static int capture_queue(Workspace& ws, hipStream_t st, const uint32_t* queue, uint32_t live, const double* rays_dev, const PathHit* hit_dev, const double* rec_dev, Capture& cap) {
    HIP_OK(hipStreamSynchronize(st));
    cap.queue_len = live;
    std::vector<uint32_t> q(live);
    HIP_OK(hipMemcpy(q.data(), queue, (size_t)live * sizeof(uint32_t), hipMemcpyDeviceToHost));
    for (uint32_t k = 0; k < live && cap.prim.size() < cap.max_n; k += cap.stride) {
        const uint32_t path = q[k];
        double r[8];
        HIP_OK(hipMemcpy(r, rays_dev + (size_t)path * RT_RAY_STRIDE, sizeof r, hipMemcpyDeviceToHost));
        for (int c = 0; c < 7; ++c) cap.rays.push_back(r[c]);
        if (hit_dev) {   // path rays: the hit record k_shade will consume (indexed by queue slot)
            PathHit ph;
            HIP_OK(hipMemcpy(&ph, hit_dev + k, sizeof ph, hipMemcpyDeviceToHost));
            cap.prim.push_back(ph.prim); cap.inst.push_back(ph.inst); cap.t.push_back(ph.t); cap.t_light.push_back(0.0);
        } else {         // shadow rays: t_light travels in the ray block, the walk's t goes to the bounce record
            double ts;
            HIP_OK(hipMemcpy(&ts, rec_dev + (size_t)path * REC_STRIDE, sizeof ts, hipMemcpyDeviceToHost));
            cap.prim.push_back(RT_NONE); cap.inst.push_back(RT_NONE); cap.t.push_back(ts); cap.t_light.push_back(r[7]);
        }
    }
    return RT_OK;
}

static int run_paths(rt_scene* sc, Workspace& ws, hipStream_t st, uint32_t n, bool explicit_lists, uint32_t width, uint32_t height, uint32_t sample0, int32_t max_depth,
                     int32_t integrator, uint64_t seed, bool count_tests, bool sort_rays, EventTimer& tm, uint64_t& launches, uint64_t* shadow_rays = nullptr, Capture* cap = nullptr,
                     bool lane_walk = false, bool want_sig = false, bool bins = false, bool count_steps = true) {
    PathBuffers pbk = ws.pb;   // what the kernels see: signatures only for rt_sample_radiance_sig
    if (!want_sig) pbk.sig = nullptr;
    else HIP_OK(hipMemsetAsync(pbk.sig, 0, (size_t)n * sizeof(unsigned long long), st));
    const bool use_walk = sc->walk_ok && !count_tests && lane_walk;   // opt-in (RT_RENDER_LANE_WALK): bounce >= 1 and shadow rays by the per-lane walk (k_walk)
    const uint32_t npix = width * height;
    tm.begin(CAT_OTHER);
    hipLaunchKernelGGL(k_raygen, grid_for(n), dim3(256), 0, st, sc->dev.camera, width, height, n, sample0, npix,
                       explicit_lists ? ws.in_pixel : nullptr, explicit_lists ? ws.in_sample : nullptr, seed, ws.pb.rays, ws.cap, ws.pb.info,
                       explicit_lists ? 0u : 1u);
    hipLaunchKernelGGL(k_iota, grid_for(n), dim3(256), 0, st, ws.queue[0], n);
    hipLaunchKernelGGL(k_set_u32, dim3(1), dim3(1), 0, st, ws.counts + 0, n);
    tm.end();
    launches += 3;
    int cur = 0;
    uint32_t live = n;  // rays in queue[cur]: read back after every shade (one 8-byte copy and a stream sync per bounce), so
                        // that sorts and launches of the thin deeper bounces are sized to the survivors, not to the pass
    for (int32_t bounce = 0; bounce < max_depth && live > 0; ++bounce) {
        const int nxt = cur ^ 1;
        HIP_OK(hipMemsetAsync(ws.counts + nxt, 0, sizeof(uint32_t), st));
        HIP_OK(hipMemsetAsync(ws.counts + 2, 0, sizeof(uint32_t), st));
        tm.begin(CAT_TRACE);
        if (use_walk && bounce >= 1) { int rc = launch_walk(sc, ws, st, ws.queue[cur], ws.counts + cur, live, ws.pb.rays, 0.001, 100000.0, ws.pb.hit, nullptr, 1u, false); if (rc != RT_OK) return rc; }
        else launch_trace(sc, ws.stats, ws.cap, st, count_tests, ws.queue[cur], ws.counts + cur, live, ws.pb.rays, 0.001, 100000.0, nullptr, nullptr, nullptr, 64u, 1u, ws.pb.hit, false, nullptr, count_steps);  // integrator.rs:38
        tm.end();
        if (cap && !cap->shadow && (uint32_t)bounce == cap->bounce) return capture_queue(ws, st, ws.queue[cur], live, ws.pb.rays, ws.pb.hit, nullptr, *cap);
        const bool mis_shadow = integrator == RT_INT_MIS && sc->dev.n_lights > 0;
        const bool last_bounce = bounce + 1 >= max_depth;
        tm.begin(CAT_SHADE);
        const uint32_t* order = nullptr;
        if (sc->bin_materials && bins) {   // opt-in (RT_RENDER_MATERIAL_BINS): regroup the queue's slots by the material class of their hits (compaction by material)
            if (!ws.order) {   // first use on this workspace (freed with it)
                int rc;
                if ((rc = ws_alloc(ws, ws.order, ws.cap)) != RT_OK || (rc = ws_alloc(ws, ws.cls, ws.cap)) != RT_OK ||
                    (rc = ws_alloc(ws, ws.bins, (size_t)RT_N_CLASSES * ((ws.cap + 511) / 512) + 16)) != RT_OK) return rc;
            }
            const uint32_t rounds = live > (8u << 20) ? 32u : 8u, nunits = (live + rounds * 64u - 1u) / (rounds * 64u);
            const dim3 ugrid((nunits + 3u) / 4u);
            hipLaunchKernelGGL(k_bin_count, ugrid, dim3(256), 0, st, sc->dev, ws.pb.hit, live, nunits, rounds, ws.cls, ws.bins);
            hipLaunchKernelGGL(k_bin_scan, dim3(1), dim3(1024), 0, st, ws.bins, RT_N_CLASSES * nunits);
            hipLaunchKernelGGL(k_bin_scatter, ugrid, dim3(256), 0, st, ws.cls, live, nunits, rounds, ws.bins, ws.order);
            order = ws.order;
            launches += 3;
        }
        // shade appends survivors (and, when sorting, their coherence keys) compactly to the raw queues
        if (sort_rays)
            hipLaunchKernelGGL(k_shade, dim3((live + RT_SHADE_BLOCK * RT_SHADE_TILES - 1u) / (RT_SHADE_BLOCK * RT_SHADE_TILES)), dim3(RT_SHADE_BLOCK), 0, st, sc->dev, pbk, ws.queue[cur], order, ws.counts + cur, (uint32_t)bounce, max_depth, integrator,
                               seed, ws.raw_queue, ws.raw_keys, ws.counts + nxt, ws.raw_squeue, ws.raw_skeys, ws.counts + 2);
        else
            hipLaunchKernelGGL(k_shade_dense, dim3((live + RT_SHADE_BLOCK * RT_SHADE_TILES - 1u) / (RT_SHADE_BLOCK * RT_SHADE_TILES)), dim3(RT_SHADE_BLOCK), 0, st, sc->dev, pbk, ws.queue[cur], order, ws.counts + cur, (uint32_t)bounce, max_depth, integrator,
                               seed, ws.queue[nxt], (uint32_t*)nullptr, ws.counts + nxt, ws.shadow_queue, (uint32_t*)nullptr, ws.counts + 2);
        HIP_OK(hipMemcpyAsync(ws.h_counts, ws.counts, 3 * sizeof(uint32_t), hipMemcpyDeviceToHost, st));
        HIP_OK(hipStreamSynchronize(st));
        const uint32_t next_live = std::min(ws.h_counts[nxt], n), shadow_live = std::min(ws.h_counts[2], n);
        if (sort_rays) {  // the sorted order lands in queue[nxt] / shadow_queue
            int rc = RT_OK;
            if (mis_shadow && shadow_live) rc = sort_queue(ws, st, shadow_live, ws.raw_skeys, ws.raw_squeue, ws.shadow_queue);
            if (rc == RT_OK && !last_bounce && next_live) rc = sort_queue(ws, st, next_live, ws.raw_keys, ws.raw_queue, ws.queue[nxt]);
            if (rc != RT_OK) return rc;
        }
        tm.end();
        launches += 2;
        if (mis_shadow && shadow_live) {
            if (shadow_rays) *shadow_rays += shadow_live;
            tm.begin(CAT_SHADOW);
            double* t_shadow = ws.pb.records + (size_t)bounce * ws.cap * REC_STRIDE + REC_T_SHADOW;
            if (use_walk) { int rc = launch_walk(sc, ws, st, ws.shadow_queue, ws.counts + 2, shadow_live, ws.pb.shadow_rays, 0.001, 10000.0, nullptr, t_shadow, (uint32_t)REC_STRIDE, true); if (rc != RT_OK) return rc; }
            else launch_trace(sc, ws.stats, ws.cap, st, count_tests, ws.shadow_queue, ws.counts + 2, shadow_live, ws.pb.shadow_rays, 0.001, 10000.0, nullptr, nullptr,
                              t_shadow, 64u, (uint32_t)REC_STRIDE, nullptr, true, nullptr, count_steps);  // integrator.rs:82
            tm.end();
            launches += 1;
            if (cap && cap->shadow && (uint32_t)bounce == cap->bounce)
                return capture_queue(ws, st, ws.shadow_queue, shadow_live, ws.pb.shadow_rays, nullptr, ws.pb.records + (size_t)bounce * ws.cap * REC_STRIDE + REC_T_SHADOW, *cap);
        }
        cur = nxt;
        live = next_live;
    }
    tm.begin(CAT_OTHER);
    hipLaunchKernelGGL(k_unwind, grid_for(n), dim3(256), 0, st, pbk, n);
    tm.end();
    launches += 1;
    HIP_OK(hipGetLastError());
    return RT_OK;
}


// ---- strict pruning: host side -----------------------------------------------------------------------------------------
// (proof and definitions: DESIGN.md section 5, "Strict pruning"; device side: StrictFilter / ray_certified)
/* RT_CERT_K (direction grid: 3 folded cube-map faces of K x K cells) and RT_CERT_EPS (a ray is certified against a normal
 * n^ iff |d.n^| >= EPS |d|_inf) are defined with ray_certified above */
#define RT_CERT_MAX_NORMALS (1u << 17)
#define RT_PRUNE_MIN_TRIS 64u
#define RT_PRUNE_MIN_SHAPE 0x1p-10        /* |e1 x e2|_2 / (|e1|_1 |e2|_1) below this: the triangle is never pruned */
struct PruneHost {
    bool on = false;
    uint32_t xf = RT_NONE;                 // the one transform space whose triangles are pruned
    double tau_coef = 0.0, radius = 0.0, box_abs_max = 0.0, shape_min = 0.0;
    uint64_t n_tris = 0, n_prunable_nodes = 0;
    std::vector<float> local_box;          // 6 per node: (lo, hi) per axis in space xf, (-inf, +inf) = never pruned
    std::vector<double> normals;           // 3 per distinct (clustered) unit normal
    std::vector<uint32_t> cell_off;        // CSR over direction cells into entries (lists padded to a multiple of 4 by repeating their last entry)
    std::vector<CertEntry> entries;        // normals whose "nearly perpendicular" band meets the cell
    std::vector<uint32_t> plane_off;       // CSR over clusters into planes
    std::vector<double> planes;            // sorted plane offsets v0 . n^ of the cluster's triangles
    std::vector<uint32_t> plane_leaf;      // parallel: node index of the triangle's leaf (prepare_scene turns it into the record offset)
    double delta_coef = 0.0;
    bool has_moving = false;               // a covered moving sphere: pruning only for rays inside the camera's shutter
    double time0 = 0.0, time1 = 0.0;
};
static float f32_up_host(double v) {       // smallest float >= v (NaN / overflow: +inf)
    if (!(v <= 3.4e38)) return std::numeric_limits<float>::infinity();
    float f = (float)v;
    if ((double)f < v) f = std::nextafterf(f, std::numeric_limits<float>::infinity());
    return f;
}
static float f32_down_host(double v) { return -f32_up_host(-v); }
template <class EMIT>
static void cert_raster(const double* n, EMIT&& emit) {
    // Face a: directions with |d_a| = |d|_inf, scaled to d_a = 1: (u, v) = (d_b, d_c) / d_a in [-1, 1]^2 and
    // d.n^ / |d|_inf = +-g(u, v), g = n_a + n_b u + n_c v (gnomonic projection: the great circle is a straight line).
    // A cell is listed when |g(centre)| <= EPS + (|n_b| + |n_c|)(half cell + 2^-30) + 2^-40: g varies by at most
    // (|n_b| + |n_c|) half over the cell, the 2^-30 covers the rounding of the cell index computed on the device.
    const uint32_t K = RT_CERT_K;
    const double cell = 2.0 / K, half = 1.0 / K;
    for (int a = 0; a < 3; ++a) {
        const int b = (a + 1) % 3, c = (a + 2) % 3;
        const double na = n[a], nb = n[b], nc = n[c];
        const double thr = RT_CERT_EPS + (std::fabs(nb) + std::fabs(nc)) * (half + 0x1p-30) + 0x1p-40;
        const bool solve_u = std::fabs(nb) >= std::fabs(nc);  // solve along the axis with the larger coefficient
        const double ns = solve_u ? nb : nc, nr = solve_u ? nc : nb;
        for (uint32_t j = 0; j < K; ++j) {
            const double r = -1.0 + ((double)j + 0.5) * cell;
            const double k = na + nr * r;
            int64_t i0 = 0, i1 = (int64_t)K - 1;
            if (std::fabs(ns) > 1e-9) {
                double lo = (-thr - k) / ns, hi = (thr - k) / ns;
                if (lo > hi) std::swap(lo, hi);
                if (hi < -1.0 - cell || lo > 1.0 + cell) continue;
                i0 = (int64_t)std::floor((std::max(lo, -1.0 - cell) + 1.0) / cell) - 1;
                i1 = (int64_t)std::floor((std::min(hi, 1.0 + cell) + 1.0) / cell) + 1;
                i0 = std::max<int64_t>(i0, 0); i1 = std::min<int64_t>(i1, (int64_t)K - 1);
            }
            for (int64_t i = i0; i <= i1; ++i) {
                const double sc = -1.0 + ((double)i + 0.5) * cell;
                if (std::fabs(k + ns * sc) > thr * (1.0 + 0x1p-40)) continue;
                const uint32_t iu = solve_u ? (uint32_t)i : j, iv = solve_u ? j : (uint32_t)i;
                emit(((uint32_t)a * K + iv) * K + iu);
            }
        }
    }
}
template <class IS_TRI2, class IS_SPH2>
static void build_prune(const rt_blob_header& h, const rt_node* nodes, const rt_geom* geom, IS_TRI2&& is_tri2, IS_SPH2&& is_sph2, PruneHost& R) {
    const float inf = std::numeric_limits<float>::infinity();
    R.local_box.assign((size_t)h.n_nodes * 6, 0.f);
    for (size_t i = 0; i < (size_t)h.n_nodes; ++i)
        for (int k = 0; k < 3; ++k) { R.local_box[i * 6 + 2 * k] = -inf; R.local_box[i * 6 + 2 * k + 1] = inf; }
    if (h.root_end <= h.root_node) return;
    {   // skip ranges of the scene-level array must nest (they do for any tree; a blob is only validated for forward links)
        std::vector<uint32_t> open;
        for (uint32_t i = h.root_node; i < h.root_end; ++i) {
            while (!open.empty() && open.back() <= i) open.pop_back();
            if (!open.empty() && nodes[i].skip > open.back()) return;
            open.push_back(nodes[i].skip);
        }
    }
    // the space with the most triangles / spheres / moving spheres
    auto coverable = [](uint32_t kind) { return kind == RT_PRIM_TRIANGLE || kind == RT_PRIM_SPHERE || kind == RT_PRIM_MOVING_SPHERE; };
    std::vector<std::pair<uint32_t, uint64_t>> per_xf;
    for (uint32_t i = h.root_node; i < h.root_end; ++i)
        for (uint32_t p : {nodes[i].prim0, nodes[i].prim1}) {
            if (p == RT_NONE || !coverable(geom[p].kind)) continue;
            const uint32_t xf = geom[p].xform;
            bool found = false;
            for (auto& e : per_xf) if (e.first == xf) { e.second += 1; found = true; break; }
            if (!found && per_xf.size() < 64) per_xf.push_back({xf, 1});
        }
    uint64_t best = 0;
    for (auto& e : per_xf) if (e.second > best) { best = e.second; R.xf = e.first; }
    if (best < RT_PRUNE_MIN_TRIS) return;
    // per leaf: can every primitive be covered, and the box of the leaf's primitives in their own space
    const uint32_t n0 = h.root_node, n1 = h.root_end;
    std::vector<uint8_t> ok((size_t)(n1 - n0), 0);
    std::vector<double> box((size_t)(n1 - n0) * 6);
    double shape_min = 1.0, e_max = 0.0, v_max = 0.0;
    struct Key { int64_t k[3]; bool operator<(const Key& o) const { return std::lexicographical_compare(k, k + 3, o.k, o.k + 3); } };
    struct Item { Key first; std::array<double, 3> second, v0; uint32_t node; };
    std::vector<Item> keyed;
    double sph_radius = 0.0;  // max over covered spheres of |centre|_inf + |radius|
    uint64_t n_sph = 0;
    // moving spheres are boxed over the camera's shutter; a ray with a time outside it is walked without pruning
    const double Ta = std::min(h.camera.time0, h.camera.time1), Tb = std::max(h.camera.time0, h.camera.time1);
    for (uint32_t i = n0; i < n1; ++i) {
        const rt_node& nd = nodes[i];
        if (nd.prim0 == RT_NONE) continue;
        bool good = true;
        double lo[3] = {INFINITY, INFINITY, INFINITY}, hi[3] = {-INFINITY, -INFINITY, -INFINITY};
        std::array<double, 3> nh[2], vz[2]; int nn = 0;
        double smin = 1.0, emax = 0.0, vmax = 0.0, rad = 0.0;
        uint32_t nsph_here = 0; bool moving_here = false;
        for (uint32_t p : {nd.prim0, nd.prim1}) {
            if (p == RT_NONE) continue;
            if (geom[p].xform != R.xf || !coverable(geom[p].kind)) { good = false; break; }
            if (geom[p].kind != RT_PRIM_TRIANGLE) {
                // Sphere (shape.rs:93-121, moving: 202-247): an accepted hit needs a computed discriminant >= 0; its error against
                // |d|^2 (r^2 - rho^2) (rho = distance of the centre from the ray's line) is below 24 u |d|^2 (|o - c|^2 + r^2), so the
                // line passes within 2^-24 (|o - c|_2 + |r|) <= 2^-23 (|o|_inf + |c|_inf + |r|) of the ball: covered by tau (tau_coef >= 2^-21).
                const bool mv = geom[p].kind == RT_PRIM_MOVING_SPHERE;
                const double r = std::fabs(geom[p].d[mv ? 8 : 3]);
                good = good && r >= 0x1p-100 && r <= 0x1p100;
                double ends[2][3];
                for (int k = 0; k < 3 && good; ++k) {
                    if (mv) {  // centre(time) = c0 + (c1 - c0) ((time - t0) / (t1 - t0)), linear: the shutter's end points bound it
                        const double c0 = geom[p].d[k], c1 = geom[p].d[3 + k], t0 = geom[p].d[6], t1 = geom[p].d[7];
                        ends[0][k] = c0 + (c1 - c0) * ((Ta - t0) / (t1 - t0)); ends[1][k] = c0 + (c1 - c0) * ((Tb - t0) / (t1 - t0));
                    } else ends[0][k] = ends[1][k] = geom[p].d[k];
                    for (int e = 0; e < 2; ++e) {
                        good = good && std::isfinite(ends[e][k]) && std::fabs(ends[e][k]) <= 0x1p100;
                        const double slack = 0x1p-40 * (std::fabs(ends[e][k]) + r);  // rounding of the centre at the ray's time
                        lo[k] = std::min(lo[k], std::nextafter(ends[e][k] - r - slack, -INFINITY)); hi[k] = std::max(hi[k], std::nextafter(ends[e][k] + r + slack, INFINITY));
                        rad = std::max(rad, std::fabs(ends[e][k]) + r);
                    }
                }
                if (!good) break;
                ++nsph_here; moving_here = moving_here || mv;
                continue;
            }
            const double* v0 = geom[p].d; const double* e1 = geom[p].d + 3; const double* e2 = geom[p].d + 6;
            for (int k = 0; k < 9; ++k) good = good && std::isfinite(geom[p].d[k]);
            const double E1 = std::fabs(e1[0]) + std::fabs(e1[1]) + std::fabs(e1[2]), E2 = std::fabs(e2[0]) + std::fabs(e2[1]) + std::fabs(e2[2]);
            const double V = std::max(std::max(std::fabs(v0[0]), std::fabs(v0[1])), std::fabs(v0[2]));
            good = good && E1 >= 0x1p-100 && E1 <= 0x1p100 && E2 >= 0x1p-100 && E2 <= 0x1p100 && V <= 0x1p100;
            if (!good) break;
            const double n[3] = {e1[1] * e2[2] - e1[2] * e2[1], e1[2] * e2[0] - e1[0] * e2[2], e1[0] * e2[1] - e1[1] * e2[0]};
            const double len = std::sqrt(n[0] * n[0] + n[1] * n[1] + n[2] * n[2]);
            const double shape = len / (E1 * E2);
            good = good && shape >= RT_PRUNE_MIN_SHAPE * (1.0 + 0x1p-20);
            if (!good) break;
            smin = std::min(smin, shape); emax = std::max(emax, std::max(E1, E2)); vmax = std::max(vmax, V);
            int big = 0;
            for (int k = 1; k < 3; ++k) if (std::fabs(n[k]) > std::fabs(n[big])) big = k;
            const double sgn = n[big] < 0.0 ? -1.0 : 1.0;
            vz[nn] = {v0[0], v0[1], v0[2]};
            nh[nn++] = {sgn * n[0] / len, sgn * n[1] / len, sgn * n[2] / len};
            for (int k = 0; k < 3; ++k) {   // vertices v0, v0 + e1, v0 + e2 (sums rounded outward)
                const double a = v0[k] + e1[k], c = v0[k] + e2[k];
                lo[k] = std::min(lo[k], std::min(v0[k], std::min(std::nextafter(a, -INFINITY), std::nextafter(c, -INFINITY))));
                hi[k] = std::max(hi[k], std::max(v0[k], std::max(std::nextafter(a, INFINITY), std::nextafter(c, INFINITY))));
            }
        }
        if (!good) continue;
        ok[i - n0] = 1;
        for (int k = 0; k < 3; ++k) { box[(size_t)(i - n0) * 6 + 2 * k] = lo[k]; box[(size_t)(i - n0) * 6 + 2 * k + 1] = hi[k]; }
        shape_min = std::min(shape_min, smin); e_max = std::max(e_max, emax); v_max = std::max(v_max, vmax);
        sph_radius = std::max(sph_radius, rad); n_sph += nsph_here; R.has_moving = R.has_moving || moving_here;
        R.n_tris += (uint64_t)nn;
        for (int t = 0; t < nn; ++t) {
            Key key;
            for (int k = 0; k < 3; ++k) key.k[k] = (int64_t)std::llround(nh[t][k] * 0x1p34);
            keyed.push_back({key, nh[t], vz[t], i});
        }
    }
    // inner nodes, bottom-up: prunable iff every leaf below is a good leaf of the space (children: i+1, skip(i+1), ...)
    for (uint32_t i = n1; i-- > n0;) {
        const rt_node& nd = nodes[i];
        if (nd.prim0 != RT_NONE) continue;
        bool good = nd.skip > i + 1;
        double lo[3] = {INFINITY, INFINITY, INFINITY}, hi[3] = {-INFINITY, -INFINITY, -INFINITY};
        for (uint32_t c = i + 1; good && c < nd.skip; c = nodes[c].skip) {
            good = ok[c - n0] != 0;
            for (int k = 0; k < 3 && good; ++k) { lo[k] = std::min(lo[k], box[(size_t)(c - n0) * 6 + 2 * k]); hi[k] = std::max(hi[k], box[(size_t)(c - n0) * 6 + 2 * k + 1]); }
        }
        if (!good) continue;
        ok[i - n0] = 1;
        for (int k = 0; k < 3; ++k) { box[(size_t)(i - n0) * 6 + 2 * k] = lo[k]; box[(size_t)(i - n0) * 6 + 2 * k + 1] = hi[k]; }
    }
    double babs = 0.0;
    for (uint32_t i = n0; i < n1; ++i) {
        if (!ok[i - n0] || nodes[i].prim0 != RT_NONE) continue;   // only inner records carry a local box
        ++R.n_prunable_nodes;
        for (int k = 0; k < 3; ++k) {
            const double lo = box[(size_t)(i - n0) * 6 + 2 * k], hi = box[(size_t)(i - n0) * 6 + 2 * k + 1];
            R.local_box[(size_t)i * 6 + 2 * k] = f32_down_host(lo); R.local_box[(size_t)i * 6 + 2 * k + 1] = f32_up_host(hi);
            babs = std::max(babs, std::max(std::fabs(lo), std::fabs(hi)));
        }
    }
    if (R.n_prunable_nodes == 0 || (keyed.empty() && n_sph == 0)) return;
    R.n_tris += n_sph;  // primitives covered
    // distinct normals: members of a cluster differ from its representative by < 2^-33 per component
    std::sort(keyed.begin(), keyed.end(), [](const auto& a, const auto& b) { return a.first < b.first; });
    // ... and, per cluster, the sorted offsets v0 . n^ of its members' planes (second stage of the certificate)
    R.plane_off.clear(); R.planes.clear(); R.plane_leaf.clear();
    {
        std::vector<std::pair<double, uint32_t>> members;
        auto flush = [&]() {
            std::sort(members.begin(), members.end());
            for (auto& m : members) { R.planes.push_back(m.first); R.plane_leaf.push_back(m.second); }
            members.clear();
        };
        for (size_t i = 0; i < keyed.size(); ++i) {
            if (i == 0 || keyed[i - 1].first < keyed[i].first) {
                flush();
                R.normals.push_back(keyed[i].second[0]); R.normals.push_back(keyed[i].second[1]); R.normals.push_back(keyed[i].second[2]);
                R.plane_off.push_back((uint32_t)R.planes.size());
            }
            const double* n = &R.normals[R.normals.size() - 3];
            members.push_back({keyed[i].v0[0] * n[0] + keyed[i].v0[1] * n[1] + keyed[i].v0[2] * n[2], keyed[i].node});
        }
        flush();
        R.plane_off.push_back((uint32_t)R.planes.size());
        if (keyed.empty()) R.plane_off.assign(1, 0u);
    }
    const size_t M = R.normals.size() / 3;
    if (M > RT_CERT_MAX_NORMALS) { R.normals.clear(); return; }
    const size_t ncell = 3u * (size_t)RT_CERT_K * RT_CERT_K;
    R.cell_off.assign(ncell + 1, 0);
    for (size_t m = 0; m < M; ++m) cert_raster(&R.normals[3 * m], [&](uint32_t c) { ++R.cell_off[c + 1]; });
    std::vector<uint32_t> fill(ncell);
    {
        uint64_t total = 0;
        for (size_t c = 0; c < ncell; ++c) {
            const uint32_t cnt = R.cell_off[c + 1];
            R.cell_off[c] = (uint32_t)total; fill[c] = (uint32_t)total;
            total += (cnt + 3u) & ~3u;
            if (total > 0x7FFFFFF0ull) { R.normals.clear(); R.cell_off.clear(); return; }
        }
        R.cell_off[ncell] = (uint32_t)total;
        CertEntry none; none.nx = none.ny = none.nz = 0.f; none.cluster = 0xFFFFFFFFu;
        R.entries.assign((size_t)total, none);
    }
    for (size_t m = 0; m < M; ++m)
        cert_raster(&R.normals[3 * m], [&](uint32_t c) {
            CertEntry& en = R.entries[fill[c]++];
            en.nx = (float)R.normals[3 * m]; en.ny = (float)R.normals[3 * m + 1]; en.nz = (float)R.normals[3 * m + 2]; en.cluster = (uint32_t)m;
        });
    for (size_t c = 0; c < ncell; ++c)  // padding repeats the list's last entry (a repeated test changes nothing)
        for (uint32_t e = fill[c]; e < R.cell_off[c + 1]; ++e) R.entries[e] = R.entries[fill[c] - 1];
    // tau = tau_coef (|o|_inf + radius) bounds the distance between the ray point the reference accepts and the triangle
    R.shape_min = shape_min;
    R.tau_coef = 0x1p-21 / shape_min;   // >= 2^-21: also covers the spheres' 2^-23
    R.delta_coef = 0x1p-20 / shape_min;
    R.radius = std::max(v_max + e_max, sph_radius);
    R.time0 = R.has_moving ? Ta : -INFINITY; R.time1 = R.has_moving ? Tb : INFINITY;
    R.box_abs_max = babs;
    R.on = true;
}


extern "C" {

const char* rt_last_error(void) { return g_err.c_str(); }

int rt_device_count(void) {
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess) return fail(RT_ERR_HIP, hipGetErrorString(e));
    return n;
}

int rt_context_create(int device, void* stream, rt_context** out) {
    if (!out) return fail(RT_ERR_BAD_ARG, "rt_context_create: out is NULL");
    int n = 0;
    HIP_OK(hipGetDeviceCount(&n));
    if (device < 0 || device >= n) return fail(RT_ERR_BAD_ARG, "rt_context_create: no such device");
    HIP_OK(hipSetDevice(device));
    rt_context* c = new (std::nothrow) rt_context();
    if (!c) return fail(RT_ERR_OOM, "rt_context_create: host allocation failed");
    c->device = device;
    {   // device address of this GPU's copy of the no-alloc scratch array (the device is current: hipSetDevice above)
        void* ptr = nullptr;
        hipError_t e = hipGetSymbolAddress(&ptr, HIP_SYMBOL(scratch));
        if (e != hipSuccess || !ptr) { delete c; return fail(RT_ERR_HIP, std::string("hipGetSymbolAddress(scratch): ") + hipGetErrorString(e)); }
        c->scratch = (unsigned char*)ptr;
    }
    if (stream) { c->stream = (hipStream_t)stream; c->own_stream = false; }
    else {
        hipError_t e = hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking);
        if (e != hipSuccess) { delete c; return fail(RT_ERR_HIP, hipGetErrorString(e)); }
        c->own_stream = true;
    }
    *out = c;
    return RT_OK;
}
int rt_context_destroy(rt_context* ctx) {
    if (!ctx) return RT_OK;
    if (ctx->live_scenes > 0) return fail(RT_ERR_BAD_ARG, "rt_context_destroy: " + std::to_string(ctx->live_scenes) + " scene(s) of this context are still alive (destroy them first)");
    (void)hipSetDevice(ctx->device);
    if (ctx->own_stream) (void)hipStreamDestroy(ctx->stream);
    delete ctx;
    return RT_OK;
}
int rt_context_synchronize(rt_context* ctx) {
    if (!ctx) return fail(RT_ERR_BAD_ARG, "null context");
    HIP_OK(hipSetDevice(ctx->device));
    HIP_OK(hipStreamSynchronize(ctx->stream));
    return RT_OK;
}

static bool range_ok(uint64_t off, uint64_t count, uint64_t elem, uint64_t total) {
    return off % 16 == 0 && off <= total && count * elem <= total - off;
}


// Everything rt_scene_upload derives from a blob without touching the device: validation of every index the kernels
// follow, and the trace stream.  Also behind rt_stream_build_host, so the CPU-only tests can check the stream.
struct ScenePrep {
    rt_blob_header h;
    bool boxes_finite = true, boxes_nested = false, reg_ray = false, has_instances = false, walk_ok = false;
    std::vector<uint8_t> uses_uv, stream;
    std::vector<uint32_t> rec_off, end_off;
    std::vector<uint8_t> stream_prod;  // production stream (see prepare_scene): the full one minus the always-entered inner nodes
    std::vector<uint32_t> rec_off_prod, end_off_prod;
    uint32_t root_off_prod = 0, n_elided = 0;
    double box_abs_max = 0.0;
    uint32_t root_off = 0;
    PruneHost prune;
};
static int prepare_scene(const void* blob, uint64_t nbytes, ScenePrep& P) {
    if (nbytes < sizeof(rt_blob_header)) return fail(RT_ERR_BAD_BLOB, "blob smaller than its header");
    rt_blob_header& h = P.h;
    memcpy(&h, blob, sizeof h);
    if (h.magic != RT_BLOB_MAGIC || h.version != RT_BLOB_VERSION) return fail(RT_ERR_BAD_BLOB, "bad blob magic/version");
    if (h.total_bytes != nbytes) return fail(RT_ERR_BAD_BLOB, "blob size mismatch");
    // operand-shape checks: every index the kernels follow is validated here, on the host
    if (!range_ok(h.off_nodes, h.n_nodes, sizeof(rt_node), nbytes) || !range_ok(h.off_prims, h.n_prims, sizeof(rt_prim), nbytes) ||
        !range_ok(h.off_geom, h.n_prims, sizeof(rt_geom), nbytes) || !range_ok(h.off_tri_attrs, h.n_tris, sizeof(rt_tri_attrs), nbytes) ||
        !range_ok(h.off_xforms, h.n_xforms, sizeof(rt_xform), nbytes) ||
        !range_ok(h.off_materials, h.n_materials, sizeof(rt_material), nbytes) || !range_ok(h.off_textures, h.n_textures, sizeof(rt_texture), nbytes) ||
        !range_ok(h.off_texels, h.n_texel_bytes, 1, nbytes) || !range_ok(h.off_lights, h.n_lights, sizeof(rt_light), nbytes))
        return fail(RT_ERR_BAD_BLOB, "table offset/size out of range");
    const uint8_t* b = (const uint8_t*)blob;
    const rt_node* nodes = (const rt_node*)(b + h.off_nodes);
    const rt_prim* prims = (const rt_prim*)(b + h.off_prims);
    const rt_geom* geom = (const rt_geom*)(b + h.off_geom);
    const rt_material* mats = (const rt_material*)(b + h.off_materials);
    const rt_texture* tex = (const rt_texture*)(b + h.off_textures);
    if (h.root_node > h.root_end || h.root_end > h.n_nodes) return fail(RT_ERR_BAD_BLOB, "root range out of bounds");
    bool& boxes_finite = P.boxes_finite;
    for (uint32_t i = 0; i < h.n_nodes; ++i) {
        for (int k = 0; k < 3 && !(nodes[i].flags & RT_NODE_NO_BOX); ++k)
            boxes_finite = boxes_finite && std::isfinite(nodes[i].bmin[k]) && std::isfinite(nodes[i].bmax[k]) && nodes[i].bmin[k] <= nodes[i].bmax[k];
        if (nodes[i].skip <= i || nodes[i].skip > h.n_nodes) return fail(RT_ERR_BAD_BLOB, "node skip link not strictly forward / out of range");
        if ((nodes[i].prim0 != RT_NONE && nodes[i].prim0 >= h.n_prims) || (nodes[i].prim1 != RT_NONE && nodes[i].prim1 >= h.n_prims))
            return fail(RT_ERR_BAD_BLOB, "node primitive index out of range");
        if (nodes[i].prim0 == RT_NONE && nodes[i].prim1 != RT_NONE) return fail(RT_ERR_BAD_BLOB, "node with prim1 but no prim0");
    }
    for (uint32_t i = 0; i < h.n_prims; ++i) {
        const rt_prim& p = prims[i];
        if (p.kind > RT_PRIM_INSTANCE) return fail(RT_ERR_BAD_BLOB, "primitive kind out of range");
        if ((p.kind == RT_PRIM_TRIANGLE && p.index >= h.n_tris) || (p.kind == RT_PRIM_INSTANCE && p.index >= h.n_nodes))
            return fail(RT_ERR_BAD_BLOB, "primitive index out of range");
        if (geom[i].kind != p.kind || geom[i].xform != p.xform) return fail(RT_ERR_BAD_BLOB, "geom record does not match its primitive");
        if (p.kind == RT_PRIM_INSTANCE && (geom[i].d[0] != (double)p.index || geom[i].d[1] != (double)p.sub_end))
            return fail(RT_ERR_BAD_BLOB, "instance geom range does not match its primitive");
        if (p.xform != RT_NONE && p.xform >= h.n_xforms) return fail(RT_ERR_BAD_BLOB, "primitive xform out of range");
        if (p.kind == RT_PRIM_INSTANCE) {
            if (p.sub_end <= p.index || p.sub_end > h.n_nodes || p.index < h.root_end) return fail(RT_ERR_BAD_BLOB, "instance sub-array out of range");
            for (uint32_t k = p.index; k < p.sub_end; ++k) {
                if (nodes[k].skip > p.sub_end) return fail(RT_ERR_BAD_BLOB, "instance skip link leaves its sub-array");
                for (uint32_t q : {nodes[k].prim0, nodes[k].prim1})
                    if (q != RT_NONE && (prims[q].kind == RT_PRIM_INSTANCE || prims[q].xform != RT_NONE))
                        return fail(RT_ERR_UNSUPPORTED, "nested instance / transformed primitive inside an instance");
            }
        } else if (p.material >= h.n_materials) return fail(RT_ERR_BAD_BLOB, "primitive material out of range");
    }
    for (uint32_t k = h.root_node; k < h.root_end; ++k)
        if (nodes[k].skip > h.root_end) return fail(RT_ERR_BAD_BLOB, "scene-level skip link leaves the scene array");
    std::vector<uint8_t>& uses_uv = P.uses_uv;
    uses_uv.assign(std::max<uint32_t>(h.n_materials, 1), 0);
    for (uint32_t i = 0; i < h.n_textures; ++i) {
        if (tex[i].kind > RT_TEX_IMAGE) return fail(RT_ERR_BAD_BLOB, "texture kind");
        if (tex[i].kind == RT_TEX_CHECKER && (tex[i].a >= h.n_textures || tex[i].b >= h.n_textures)) return fail(RT_ERR_BAD_BLOB, "checker child out of range");
        if (tex[i].kind == RT_TEX_IMAGE && (tex[i].b == 0 || tex[i].c == 0 || (uint64_t)tex[i].a + (uint64_t)tex[i].b * tex[i].c * 3 > h.n_texel_bytes))
            return fail(RT_ERR_BAD_BLOB, "image texture outside the texel pool");
    }
    for (uint32_t i = 0; i < h.n_materials; ++i) {
        if (mats[i].kind > RT_MAT_DIFFUSE_LIGHT) return fail(RT_ERR_BAD_BLOB, "material kind");
        bool needs_tex = mats[i].kind == RT_MAT_LAMBERTIAN || mats[i].kind == RT_MAT_DIFFUSE_LIGHT;
        if (needs_tex && mats[i].texture >= h.n_textures) return fail(RT_ERR_BAD_BLOB, "material texture out of range");
        if (needs_tex) {  // does the texture tree reach an image? (bounded walk; checker trees deeper than 8 are rejected)
            std::vector<std::pair<uint32_t, int>> stack{{mats[i].texture, 0}};
            while (!stack.empty()) {
                auto [t, dpt] = stack.back(); stack.pop_back();
                if (tex[t].kind == RT_TEX_IMAGE) uses_uv[i] = 1;
                else if (tex[t].kind == RT_TEX_CHECKER) {
                    if (dpt >= 8) return fail(RT_ERR_UNSUPPORTED, "checker texture nesting deeper than 8");
                    stack.push_back({tex[t].a, dpt + 1}); stack.push_back({tex[t].b, dpt + 1});
                }
            }
        }
    }
    const rt_light* lights = (const rt_light*)(b + h.off_lights);
    for (uint32_t i = 0; i < h.n_lights; ++i)
        if (lights[i].material >= h.n_materials) return fail(RT_ERR_BAD_BLOB, "light material out of range");

    for (uint32_t i = 0; i < h.n_prims; ++i) P.has_instances = P.has_instances || prims[i].kind == RT_PRIM_INSTANCE;
    {   // do the boxes of the scene-level array nest?  (children of node i: i + 1, then along the skip links up to skip(i))
        bool nested = boxes_finite;
        for (uint32_t i = h.root_node; i < h.root_end && nested; ++i) {
            const rt_node& a = nodes[i];
            if (a.flags & RT_NODE_NO_BOX) { nested = false; break; }
            if (a.prim0 != RT_NONE) continue;
            for (uint32_t c = i + 1; c < a.skip && nested; c = nodes[c].skip) {
                if (nodes[c].skip > a.skip) { nested = false; break; }
                for (int k = 0; k < 3; ++k) nested = nested && a.bmin[k] <= nodes[c].bmin[k] && nodes[c].bmax[k] <= a.bmax[k];
            }
        }
        P.boxes_nested = nested;
    }
    // trace stream: pre-order fp32 filter records (see k_trace)
    if (h.n_nodes >= RT_STREAM_MAX_NODES) { return fail(RT_ERR_UNSUPPORTED, "more than 2^24 BVH nodes"); }
    auto is_tri2 = [&](const rt_node& nd) {  // one triangle, or two under the same transform: leaf with a filter block
        if (nd.prim0 == RT_NONE || geom[nd.prim0].kind != RT_PRIM_TRIANGLE) return false;
        return nd.prim1 == RT_NONE || (geom[nd.prim1].kind == RT_PRIM_TRIANGLE && geom[nd.prim0].xform == geom[nd.prim1].xform);
    };
    auto is_sph2 = [&](const rt_node& nd) {  // one sphere, or two under the same transform
        if (nd.prim0 == RT_NONE || geom[nd.prim0].kind != RT_PRIM_SPHERE) return false;
        return nd.prim1 == RT_NONE || (geom[nd.prim1].kind == RT_PRIM_SPHERE && geom[nd.prim0].xform == geom[nd.prim1].xform);
    };
    build_prune(h, nodes, geom, is_tri2, is_sph2, P.prune);
    // ranges: the scene-level array and every instance sub-array (validated above: disjoint, links stay inside);
    // range_end[i] = end index of the range node i belongs to, ends_here[e] = some range ends before node e
    std::vector<uint32_t> range_end((size_t)h.n_nodes, h.n_nodes);
    std::vector<uint8_t> ends_here((size_t)h.n_nodes + 1, 0);
    for (uint32_t k = h.root_node; k < h.root_end; ++k) range_end[k] = h.root_end;
    ends_here[h.root_end] = 1;
    for (uint32_t i = 0; i < h.n_prims; ++i)
        if (prims[i].kind == RT_PRIM_INSTANCE) {
            for (uint32_t k = prims[i].index; k < prims[i].sub_end; ++k) range_end[k] = prims[i].sub_end;
            ends_here[prims[i].sub_end] = 1;
        }
    auto f32_up = [](double v) {  // smallest float >= v (NaN / overflow: +inf, which disables the filter for that record)
        if (!(v <= 3.4e38)) return std::numeric_limits<float>::infinity();
        float f = (float)v;
        if ((double)f < v) f = std::nextafterf(f, std::numeric_limits<float>::infinity());
        return f;
    };
    double& box_abs_max = P.box_abs_max;
    // Lays the records out and fills them; `elide` marks inner nodes that get NO record (their children follow directly and
    // links to them land on their first child): see the production stream below.
    auto emit = [&](const std::vector<uint8_t>& elide, std::vector<uint8_t>& stream, std::vector<uint32_t>& rec_off, std::vector<uint32_t>& end_off) -> int {
    rec_off.assign((size_t)h.n_nodes + 1, 0); end_off.assign((size_t)h.n_nodes + 1, 0);
    uint64_t stream_bytes = 0;
    for (uint32_t i = 0; i <= h.n_nodes; ++i) {
        if (ends_here[i]) { end_off[i] = (uint32_t)stream_bytes; stream_bytes += RT_STREAM_REC; }  // END record of the range(s) ending here
        rec_off[i] = (uint32_t)stream_bytes;
        if (i < h.n_nodes && !elide[i]) stream_bytes += is_tri2(nodes[i]) ? RT_STREAM_REC_TRI2 : is_sph2(nodes[i]) ? RT_STREAM_REC_SPH2 : nodes[i].prim0 != RT_NONE ? RT_STREAM_REC_LEAF : RT_STREAM_REC_INNER;
        if (stream_bytes + RT_STREAM_PAD > 0xFFFFFFF0ull) { return fail(RT_ERR_UNSUPPORTED, "trace stream larger than 4 GiB"); }
    }
    // a link to the end of the node's own range lands on that range's END record
    auto link_off = [&](uint32_t from, uint32_t to) { return to == range_end[from] ? end_off[to] : rec_off[to]; };
    stream.assign((size_t)stream_bytes + RT_STREAM_PAD, 0);
    for (uint32_t e = 0; e <= h.n_nodes; ++e) {
        if (!ends_here[e]) continue;
        const float inf = std::numeric_limits<float>::infinity();
        const float box[6] = {-inf, inf, -inf, inf, -inf, inf};
        const uint32_t words[2] = {end_off[e], RT_NODE_NO_BOX | RT_STREAM_LEAF | ((RT_STREAM_REC / 16u) << 4) | (RT_STREAM_END_NODE << 8)};
        memcpy(stream.data() + end_off[e], box, 24); memcpy(stream.data() + end_off[e] + 24, words, 8);
    }
    for (uint32_t i = 0; i < h.n_nodes; ++i) {
        if (elide[i]) continue;
        const rt_node& nd = nodes[i];
        uint8_t* rec = stream.data() + rec_off[i];
        const bool nb = (nd.flags & RT_NODE_NO_BOX) != 0, tri2 = is_tri2(nd), sph2 = is_sph2(nd);
        uint32_t fl = nd.flags & (RT_NODE_NO_BOX | RT_NODE_DUP_RIGHT);
        fl |= tri2 ? RT_STREAM_TRI2 : sph2 ? RT_STREAM_SPH2 : nd.prim0 != RT_NONE ? RT_STREAM_LEAF : RT_STREAM_INNER;
        const uint32_t rec_bytes = tri2 ? RT_STREAM_REC_TRI2 : sph2 ? RT_STREAM_REC_SPH2 : nd.prim0 != RT_NONE ? RT_STREAM_REC_LEAF : RT_STREAM_REC_INNER;
        float box[6];
        for (int k = 0; k < 3; ++k) {  // list items have no box in the reference: see the box filter
            box[2 * k] = nb ? -std::numeric_limits<float>::infinity() : (float)nd.bmin[k];
            box[2 * k + 1] = nb ? std::numeric_limits<float>::infinity() : (float)nd.bmax[k];
            if (!nb && std::isfinite(nd.bmin[k]) && std::isfinite(nd.bmax[k])) box_abs_max = std::max(box_abs_max, std::max(std::fabs(nd.bmin[k]), std::fabs(nd.bmax[k])));
        }
        const uint32_t words[2] = {link_off(i, nd.skip), fl | ((rec_bytes / 16u) << 4) | (i << 8)};
        memcpy(rec, box, 24); memcpy(rec + 24, words, 8);
        if (nd.prim0 == RT_NONE) memcpy(rec + RT_STREAM_REC, &P.prune.local_box[(size_t)i * 6], 24);  // inner: box of the subtree's triangles in their own space
        if (!tri2 && !sph2 && nd.prim0 != RT_NONE) {
            const uint32_t ids[4] = {nd.prim0, nd.prim1, geom[nd.prim0].xform, 0u};
            memcpy(rec + RT_STREAM_REC, ids, 16);
        }
        if (sph2) {
            const uint32_t head[2] = {geom[nd.prim0].xform, nd.prim1 != RT_NONE ? 2u : 1u};
            memcpy(rec + RT_STREAM_REC, head, 8);
            memcpy(rec + RT_STREAM_REC + 8, geom[nd.prim0].d, 32);  // centre xyz, radius
            if (nd.prim1 != RT_NONE) memcpy(rec + RT_STREAM_REC + 40, geom[nd.prim1].d, 32);
        }
        if (!tri2) continue;
        float blk[28] = {};
        const uint32_t xf = geom[nd.prim0].xform;
        memcpy(&blk[0], &xf, 4);
        for (int t = 0; t < 2; ++t) {  // pairs (A, B): e2, ml = v0 x e2, -n = -(e1 x e2), then the error weights (see the triangle filter)
            const rt_geom& g = geom[(t == 0 || nd.prim1 == RT_NONE) ? nd.prim0 : nd.prim1];  // a single triangle fills both halves
            const double* v0 = g.d; const double* e1 = g.d + 3; const double* e2 = g.d + 6;
            const double ml[3] = {v0[1] * e2[2] - v0[2] * e2[1], v0[2] * e2[0] - v0[0] * e2[2], v0[0] * e2[1] - v0[1] * e2[0]};
            const double nn[3] = {e1[1] * e2[2] - e1[2] * e2[1], e1[2] * e2[0] - e1[0] * e2[2], e1[0] * e2[1] - e1[1] * e2[0]};
            auto n1 = [](const double* v) { return std::fabs(v[0]) + std::fabs(v[1]) + std::fabs(v[2]); };
            auto ninf = [](const double* v) { return std::max(std::max(std::fabs(v[0]), std::fabs(v[1])), std::fabs(v[2])); };
            for (int k = 0; k < 3; ++k) {
                blk[2 + 2 * k + t] = (float)e2[k];
                blk[2 + 2 * (3 + k) + t] = (float)ml[k];
                blk[2 + 2 * (6 + k) + t] = (float)-nn[k];
            }
            blk[2 + 2 * 9 + t] = f32_up(0x1p-20 * ninf(e2));
            blk[2 + 2 * 10 + t] = f32_up(0x1p-20 * n1(ml) + 0x1p-46 * ninf(v0) * n1(e2));
            blk[2 + 2 * 11 + t] = f32_up((1.0 + 0x1p-19) * (0x1p-20 * n1(nn) + 0x1p-46 * n1(e1) * n1(e2)));
        }
        memcpy(rec + RT_STREAM_REC, blk, sizeof blk);
    }
    return RT_OK;
    };
    // Full stream: every node has a record (the counting kernel and NaN-capable waves walk it).
    std::vector<uint8_t> elide((size_t)h.n_nodes, 0);
    { int rc = emit(elide, P.stream, P.rec_off, P.end_off); if (rc != RT_OK) return rc; }
    P.root_off = h.root_node == h.root_end ? P.end_off[h.root_end] : P.rec_off[h.root_node];
    // Production stream: inner nodes that can never be pruned and whose box holds everything that can be (the ancestors of a
    // 1000-unit ground sphere) are entered by practically every ray -- they get no record.  Sound when boxes nest: a ray
    // only ever visits MORE than the reference this way, and every leaf re-decides with the reference's test on its own box;
    // a leaf box that passes with the current best implies that each ancestor box (which contains it) passed with its larger
    // best, because the slab values are monotone in the plane coordinates.
    if (P.prune.on && P.boxes_nested) {
        double core_lo[3] = {INFINITY, INFINITY, INFINITY}, core_hi[3] = {-INFINITY, -INFINITY, -INFINITY};
        for (uint32_t i = h.root_node; i < h.root_end; ++i)
            if (nodes[i].prim0 == RT_NONE && std::isfinite(P.prune.local_box[(size_t)i * 6]))
                for (int k = 0; k < 3; ++k) { core_lo[k] = std::min(core_lo[k], nodes[i].bmin[k]); core_hi[k] = std::max(core_hi[k], nodes[i].bmax[k]); }
        for (uint32_t i = h.root_node; i < h.root_end; ++i) {
            if (nodes[i].prim0 != RT_NONE || std::isfinite(P.prune.local_box[(size_t)i * 6]) || !(nodes[i].skip > i + 1)) continue;
            bool holds = true;
            for (int k = 0; k < 3; ++k) holds = holds && nodes[i].bmin[k] <= core_lo[k] && core_hi[k] <= nodes[i].bmax[k];
            if (holds) { elide[i] = 1; ++P.n_elided; }
        }
    }
    { int rc = emit(elide, P.stream_prod, P.rec_off_prod, P.end_off_prod); if (rc != RT_OK) return rc; }
    P.root_off_prod = h.root_node == h.root_end ? P.end_off_prod[h.root_end] : P.rec_off_prod[h.root_node];
    static_assert(RT_STREAM_REC + 28 * 4 == RT_STREAM_REC_TRI2, "TRI2 record layout");
    {
        uint64_t n_leaf = 0, n_tri_leaf = 0;
        for (uint32_t i = 0; i < h.n_nodes; ++i) { n_leaf += nodes[i].prim0 != RT_NONE; n_tri_leaf += is_tri2(nodes[i]); }
        P.reg_ray = 2 * n_tri_leaf < n_leaf;
    }
    {   // k_walk (per-lane walk) serves scenes with a pruned space, no BVHNode instances, and every triangle leaf in that space
        bool ok = P.prune.on && !P.has_instances && P.boxes_finite && h.root_end > h.root_node;
        for (uint32_t i = h.root_node; i < h.root_end && ok; ++i)
            if (is_tri2(nodes[i]) && geom[nodes[i].prim0].xform != P.prune.xf) ok = false;
        P.walk_ok = ok;
    }
    return RT_OK;
}

int rt_stream_build_host(const void* blob, uint64_t nbytes, void* stream_out, uint64_t capacity, uint64_t* stream_bytes, uint32_t* root_off, uint32_t* node_off) {
    if (!blob || !stream_bytes) return fail(RT_ERR_BAD_ARG, "rt_stream_build_host: null argument");
    ScenePrep P;
    int rc = prepare_scene(blob, nbytes, P);
    if (rc != RT_OK) return rc;
    *stream_bytes = P.stream.size();
    if (root_off) *root_off = P.root_off;
    if (stream_out) {
        if (capacity < P.stream.size()) return fail(RT_ERR_BAD_ARG, "rt_stream_build_host: buffer too small");
        memcpy(stream_out, P.stream.data(), P.stream.size());
    }
    if (node_off) memcpy(node_off, P.rec_off.data(), P.rec_off.size() * sizeof(uint32_t));
    return RT_OK;
}

int rt_stream_build_prod_host(const void* blob, uint64_t nbytes, void* stream_out, uint64_t capacity, uint64_t* stream_bytes, uint32_t* root_off, uint32_t* node_off) {
    if (!blob || !stream_bytes) return fail(RT_ERR_BAD_ARG, "rt_stream_build_prod_host: null argument");
    ScenePrep P;
    int rc = prepare_scene(blob, nbytes, P);
    if (rc != RT_OK) return rc;
    *stream_bytes = P.stream_prod.size();
    if (root_off) *root_off = P.root_off_prod;
    if (stream_out) {
        if (capacity < P.stream_prod.size()) return fail(RT_ERR_BAD_ARG, "rt_stream_build_prod_host: buffer too small");
        memcpy(stream_out, P.stream_prod.data(), P.stream_prod.size());
    }
    if (node_off) memcpy(node_off, P.rec_off_prod.data(), P.rec_off_prod.size() * sizeof(uint32_t));
    return RT_OK;
}

int rt_prune_info_host(const void* blob, uint64_t nbytes, double* info, uint32_t n_rays, const double* rays, uint8_t* certified) {
    if (!blob || !info) return fail(RT_ERR_BAD_ARG, "rt_prune_info_host: null argument");
    ScenePrep P;
    int rc = prepare_scene(blob, nbytes, P);
    if (rc != RT_OK) return rc;
    const PruneHost& R = P.prune;
    info[0] = R.on ? 1.0 : 0.0; info[1] = (double)R.xf; info[2] = (double)R.n_tris; info[3] = (double)R.n_prunable_nodes;
    info[4] = (double)(R.normals.size() / 3); info[5] = (double)R.entries.size(); info[6] = R.shape_min; info[7] = R.tau_coef;
    info[8] = R.delta_coef; info[9] = R.radius; info[10] = R.box_abs_max; info[11] = (double)P.n_elided;
    if (n_rays && (!rays || !certified)) return fail(RT_ERR_BAD_ARG, "rt_prune_info_host: null ray arguments");
    // the device's certificate, restated on the host for rays (o, d) ALREADY in the pruned space (tests compare it with a brute-force scan)
    for (uint32_t i = 0; i < n_rays; ++i) {
        const double* o = rays + 6 * (size_t)i; const double* d = o + 3;
        certified[i] = 0;
        if (!R.on) continue;
        const double ax = std::fabs(d[0]), ay = std::fabs(d[1]), az = std::fabs(d[2]);
        const double D = std::max(std::max(ax, ay), az), O = std::max(std::max(std::fabs(o[0]), std::fabs(o[1])), std::fabs(o[2]));
        if (!(D >= 0x1p-100 && D <= 0x1p100 && O <= 0x1p100)) continue;
        const int a = ax >= ay ? (ax >= az ? 0 : 2) : (ay >= az ? 1 : 2);
        const double u = d[(a + 1) % 3] / d[a], v = d[(a + 2) % 3] / d[a];
        const double kk = (double)RT_CERT_K * 0.5, top = (double)(RT_CERT_K - 1u);
        const uint32_t iu = (uint32_t)std::fmin(std::fmax(std::floor((u + 1.0) * kk), 0.0), top), iv = (uint32_t)std::fmin(std::fmax(std::floor((v + 1.0) * kk), 0.0), top);
        const uint32_t cell = ((uint32_t)a * RT_CERT_K + iv) * RT_CERT_K + iu;
        const double thr = RT_CERT_EPS * D, delta = R.delta_coef * (O + R.radius);
        bool ok = true;
        std::vector<uint32_t> sus;
        for (uint32_t e = R.cell_off[cell]; e < R.cell_off[cell + 1]; ++e) {
            const uint32_t c = R.entries[e].cluster;
            const double* n = &R.normals[3 * (size_t)c];
            if (std::fabs(d[0] * n[0] + d[1] * n[1] + d[2] * n[2]) >= thr) continue;
            const double q = o[0] * n[0] + o[1] * n[1] + o[2] * n[2];
            auto b = R.planes.begin() + R.plane_off[c], e2 = R.planes.begin() + R.plane_off[c + 1];
            for (auto it = std::lower_bound(b, e2, q - delta); it != e2 && !(*it > q + delta); ++it) {
                const uint32_t leaf = R.plane_leaf[(size_t)(it - R.planes.begin())];
                if (std::find(sus.begin(), sus.end(), leaf) == sus.end()) sus.push_back(leaf);
            }
        }
        ok = sus.size() <= RT_SUSPECT_SLOTS;
        certified[i] = !ok ? 0 : sus.empty() ? 1 : 2;
    }
    return RT_OK;
}

int rt_scene_upload(rt_context* ctx, const void* blob, uint64_t nbytes, rt_scene** out) {
    if (!ctx || !blob || !out) return fail(RT_ERR_BAD_ARG, "rt_scene_upload: null argument");
    ScenePrep P;
    {
        int rc = prepare_scene(blob, nbytes, P);
        if (rc != RT_OK) return rc;
    }
    const rt_blob_header& h = P.h;
    const uint8_t* b = (const uint8_t*)blob;
    const rt_node* nodes = (const rt_node*)(b + h.off_nodes);
    const rt_prim* prims = (const rt_prim*)(b + h.off_prims);
    const std::vector<uint8_t>& uses_uv = P.uses_uv; const std::vector<uint8_t>& stream = P.stream;
    const std::vector<uint32_t>& rec_off = P.rec_off;
    const bool boxes_finite = P.boxes_finite; const double box_abs_max = P.box_abs_max;
    HIP_OK(hipSetDevice(ctx->device));
    rt_scene* sc = new (std::nothrow) rt_scene();
    if (!sc) return fail(RT_ERR_OOM, "host allocation failed");
    sc->ctx = ctx; sc->hdr = h;
    ctx->live_scenes += 1;
    sc->host_prims.assign(prims, prims + h.n_prims);
    sc->has_instances = P.has_instances; sc->reg_ray = P.reg_ray;
    {   // kinds of scattering material in the scene (DiffuseLight terminates a path; it is a class of its own but no reason to bin)
        const rt_material* mats = (const rt_material*)(b + h.off_materials);
        uint32_t kinds = 0;
        for (uint32_t i = 0; i < h.n_materials; ++i) if (mats[i].kind != RT_MAT_DIFFUSE_LIGHT) kinds |= 1u << mats[i].kind;
        sc->bin_materials = __builtin_popcount(kinds) > 1;
    }
#ifndef RT_TRACE_DIAG   /* the diagnostic build reports k_trace's own step counts */
    sc->walk_ok = P.walk_ok;
#endif
    { hipDeviceProp_t prop; if (hipGetDeviceProperties(&prop, ctx->device) == hipSuccess && prop.multiProcessorCount > 0) sc->n_cu = prop.multiProcessorCount; }
    hipError_t e = hipMalloc(&sc->dblob, nbytes);
    if (e == hipSuccess) e = hipMalloc((void**)&sc->d_mat_uses_uv, uses_uv.size());
    if (e == hipSuccess) e = hipMalloc((void**)&sc->d_stream, P.stream_prod.size());
    if (e == hipSuccess) e = hipMemcpy(sc->d_stream, P.stream_prod.data(), P.stream_prod.size(), hipMemcpyHostToDevice);
    if (e == hipSuccess) e = hipMalloc((void**)&sc->d_node_off, P.rec_off_prod.size() * sizeof(uint32_t));
    if (e == hipSuccess) e = hipMemcpy(sc->d_node_off, P.rec_off_prod.data(), P.rec_off_prod.size() * sizeof(uint32_t), hipMemcpyHostToDevice);
    if (e == hipSuccess) e = hipMalloc((void**)&sc->d_stream_full, stream.size());
    if (e == hipSuccess) e = hipMemcpy(sc->d_stream_full, stream.data(), stream.size(), hipMemcpyHostToDevice);
    if (e == hipSuccess) e = hipMalloc((void**)&sc->d_node_off_full, rec_off.size() * sizeof(uint32_t));
    if (e == hipSuccess) e = hipMemcpy(sc->d_node_off_full, rec_off.data(), rec_off.size() * sizeof(uint32_t), hipMemcpyHostToDevice);
    if (e == hipSuccess) e = hipMemcpy(sc->dblob, blob, nbytes, hipMemcpyHostToDevice);
    if (e == hipSuccess) e = hipMemcpy(sc->d_mat_uses_uv, uses_uv.data(), uses_uv.size(), hipMemcpyHostToDevice);
    const PruneHost& pr = P.prune;
    if (pr.on) {
        if (e == hipSuccess) e = hipMalloc((void**)&sc->d_cert_cells, pr.cell_off.size() * sizeof(uint32_t));
        if (e == hipSuccess) e = hipMemcpy(sc->d_cert_cells, pr.cell_off.data(), pr.cell_off.size() * sizeof(uint32_t), hipMemcpyHostToDevice);
        if (e == hipSuccess) e = hipMalloc((void**)&sc->d_cert_entries, std::max<size_t>(pr.entries.size(), 1) * sizeof(CertEntry));
        if (e == hipSuccess) e = hipMemcpy(sc->d_cert_entries, pr.entries.data(), pr.entries.size() * sizeof(CertEntry), hipMemcpyHostToDevice);
        if (e == hipSuccess) e = hipMalloc((void**)&sc->d_cert_normals, std::max<size_t>(pr.normals.size(), 1) * sizeof(double));
        if (e == hipSuccess) e = hipMemcpy(sc->d_cert_normals, pr.normals.data(), pr.normals.size() * sizeof(double), hipMemcpyHostToDevice);
        if (e == hipSuccess) e = hipMalloc((void**)&sc->d_cert_planes, std::max<size_t>(pr.planes.size(), 1) * sizeof(double));
        if (e == hipSuccess) e = hipMemcpy(sc->d_cert_planes, pr.planes.data(), pr.planes.size() * sizeof(double), hipMemcpyHostToDevice);
        if (e == hipSuccess) e = hipMalloc((void**)&sc->d_cert_plane_off, pr.plane_off.size() * sizeof(uint32_t));
        if (e == hipSuccess) e = hipMemcpy(sc->d_cert_plane_off, pr.plane_off.data(), pr.plane_off.size() * sizeof(uint32_t), hipMemcpyHostToDevice);
        if (e == hipSuccess) e = hipMalloc((void**)&sc->d_cert_plane_leaf, std::max<size_t>(pr.plane_leaf.size(), 1) * sizeof(uint32_t));
        if (e == hipSuccess) e = hipMemcpy(sc->d_cert_plane_leaf, pr.plane_leaf.data(), pr.plane_leaf.size() * sizeof(uint32_t), hipMemcpyHostToDevice);
    }
    if (e != hipSuccess) {
        if (sc->dblob) (void)hipFree(sc->dblob);
        if (sc->d_mat_uses_uv) (void)hipFree(sc->d_mat_uses_uv);
        if (sc->d_stream) (void)hipFree(sc->d_stream);
        if (sc->d_node_off) (void)hipFree(sc->d_node_off);
        if (sc->d_stream_full) (void)hipFree(sc->d_stream_full);
        if (sc->d_node_off_full) (void)hipFree(sc->d_node_off_full);
        if (sc->d_cert_cells) (void)hipFree(sc->d_cert_cells);
        if (sc->d_cert_entries) (void)hipFree(sc->d_cert_entries);
        if (sc->d_cert_normals) (void)hipFree(sc->d_cert_normals);
        if (sc->d_cert_planes) (void)hipFree(sc->d_cert_planes);
        if (sc->d_cert_plane_off) (void)hipFree(sc->d_cert_plane_off);
        if (sc->d_cert_plane_leaf) (void)hipFree(sc->d_cert_plane_leaf);
        ctx->live_scenes -= 1;
        delete sc;
        return fail(e == hipErrorOutOfMemory ? RT_ERR_OOM : RT_ERR_HIP, hipGetErrorString(e));
    }
    const uint8_t* d = (const uint8_t*)sc->dblob;
    SceneDev& v = sc->dev;
    v.nodes = (const rt_node*)(d + h.off_nodes); v.prims = (const rt_prim*)(d + h.off_prims);
    v.geom = (const rt_geom*)(d + h.off_geom); v.tri_attrs = (const rt_tri_attrs*)(d + h.off_tri_attrs);
    v.xforms = (const rt_xform*)(d + h.off_xforms);
    v.materials = (const rt_material*)(d + h.off_materials); v.textures = (const rt_texture*)(d + h.off_textures);
    v.texels = d + h.off_texels; v.lights = (const rt_light*)(d + h.off_lights);
    v.mat_uses_uv = sc->d_mat_uses_uv;
    v.stream = sc->d_stream; v.root_off = P.root_off_prod;
    v.end_off = P.end_off_prod[h.root_end]; v.end_off_full = P.end_off[h.root_end];
    v.stream_full = sc->d_stream_full; v.node_off_full = sc->d_node_off_full; v.root_off_full = P.root_off;
    v.node_off = sc->d_node_off; v.box_abs_max = box_abs_max;
    v.cert_cells = sc->d_cert_cells; v.cert_entries = sc->d_cert_entries; v.cert_normals = sc->d_cert_normals;
    v.cert_planes = sc->d_cert_planes; v.cert_plane_off = sc->d_cert_plane_off; v.cert_plane_leaf = sc->d_cert_plane_leaf; v.cert_n_normals = (uint32_t)(pr.normals.size() / 3); v.prune_time0 = pr.time0; v.prune_time1 = pr.time1; v.prune_delta_coef = pr.delta_coef;
    v.prune_on = pr.on ? 1u : 0u; v.prune_xf = pr.xf;
    v.translate_xf = RT_NONE;
    {   // apply_xform<true>'s shortcut: the pruned space's transform if it qualifies, else the first that does
        const rt_xform* xf = (const rt_xform*)((const uint8_t*)blob + h.off_xforms);
        auto pure_translation = [&](uint32_t i) {
            const double* m = xf[i].t_inv;
            for (int c = 0; c < 3; ++c)
                for (int r = 0; r < 3; ++r) if (m[4 * c + r] != (r == c ? 1.0 : 0.0)) return false;
            for (int r = 0; r < 3; ++r) if (!std::isfinite(m[12 + r]) || m[12 + r] == 0.0) return false;
            return true;
        };
        if (pr.on && pr.xf != RT_NONE && pr.xf < h.n_xforms && pure_translation(pr.xf)) v.translate_xf = pr.xf;
        else for (uint32_t i = 0; i < h.n_xforms && v.translate_xf == RT_NONE; ++i) if (pure_translation(i)) v.translate_xf = i;
    }
#ifdef RT_NO_TRANSLATE_SHORTCUT   /* A/B */
    v.translate_xf = RT_NONE;
#endif
    v.prune_tau_coef = pr.tau_coef; v.prune_radius = pr.radius; v.prune_box_abs_max = pr.box_abs_max;
    v.n_nodes = h.n_nodes; v.boxes_finite = boxes_finite ? 1u : 0u;
    v.camera = h.camera; v.n_lights = h.n_lights; v.root_node = h.root_node; v.root_end = h.root_end; v.background = h.background;
    {   // ray-sorting grid: mean +- 2.5 sigma of the scene-level leaf-box centres (robust against a 1000-unit ground sphere
        // or far-away light spheres); origins outside clamp to the border cells
        double sum[3] = {0, 0, 0}, sq[3] = {0, 0, 0};
        uint64_t cnt = 0;
        for (uint32_t i = h.root_node; i < h.root_end; ++i) {
            const rt_node& nd = nodes[i];
            if ((nd.flags & RT_NODE_NO_BOX) || nd.prim0 == RT_NONE) continue;
            bool ok = true;
            for (int k = 0; k < 3; ++k) ok = ok && std::isfinite(nd.bmin[k]) && std::isfinite(nd.bmax[k]) && (nd.bmax[k] - nd.bmin[k]) <= 500.0;
            if (!ok) continue;
            for (int k = 0; k < 3; ++k) { double c = 0.5 * (nd.bmin[k] + nd.bmax[k]); sum[k] += c; sq[k] += c * c; }
            ++cnt;
        }
        sc->sortable = cnt >= 64;
        for (int k = 0; k < 3; ++k) {
            double mean = cnt ? sum[k] / (double)cnt : 0.0;
            double sd = cnt ? std::sqrt(std::max(sq[k] / (double)cnt - mean * mean, 0.0)) : 1.0;
            double ext = std::max(5.0 * sd, 1e-6);
            v.sort_min[k] = (float)(mean - 0.5 * ext);
            v.sort_scale[k] = (float)(32.0 / ext);
        }
    }
    *out = sc;
    return RT_OK;
}

int rt_render_end(rt_job* job, rt_stats* stats);
int rt_scene_destroy(rt_scene* sc) {
    if (!sc) return RT_OK;
    (void)hipSetDevice(sc->ctx->device);
    for (int k = 0; k < 2; ++k) {   // jobs nobody ended: in the order they were begun
        rt_job* first = sc->lane_job[0] && (!sc->lane_job[1] || sc->lane_job[0]->seq < sc->lane_job[1]->seq) ? sc->lane_job[0] : sc->lane_job[1];
        if (first) (void)rt_render_end(first, nullptr);
    }
    { std::lock_guard<std::mutex> lk(sc->job_mu); sc->workers_quit = true; }
    sc->job_cv.notify_all();
    for (auto& w : sc->worker) if (w.joinable()) w.join();
    for (auto& e : sc->job_done) if (e) (void)hipEventDestroy(e);
    (void)hipStreamSynchronize(sc->ctx->stream);
    ws_free(sc->ws); ws_free(sc->ws2);
    if (sc->stream2) (void)hipStreamDestroy(sc->stream2);
    (void)hipFree(sc->dblob);
    (void)hipFree(sc->d_mat_uses_uv);
    (void)hipFree(sc->d_stream);
    (void)hipFree(sc->d_node_off);
    (void)hipFree(sc->d_stream_full);
    (void)hipFree(sc->d_node_off_full);
    if (sc->d_cert_cells) (void)hipFree(sc->d_cert_cells);
    if (sc->d_cert_entries) (void)hipFree(sc->d_cert_entries);
    if (sc->d_cert_normals) (void)hipFree(sc->d_cert_normals);
    if (sc->d_cert_planes) (void)hipFree(sc->d_cert_planes);
    if (sc->d_cert_plane_off) (void)hipFree(sc->d_cert_plane_off);
    if (sc->d_cert_plane_leaf) (void)hipFree(sc->d_cert_plane_leaf);
    sc->ctx->live_scenes -= 1;
    delete sc;
    return RT_OK;
}

int rt_trace_closest_device(rt_scene* sc, uint32_t n, const double* rays_dev, double t_min, double t_max, uint32_t* prim_id_dev, double* t_dev) {
    if (!sc || (n && (!rays_dev || !prim_id_dev || !t_dev))) return fail(RT_ERR_BAD_ARG, "rt_trace_closest_device: null argument");
    if (n == 0) return RT_OK;
    RT_NOT_BUSY(sc, "rt_trace_closest_device");
    HIP_OK(hipSetDevice(sc->ctx->device));
    int rc = ws_ensure(sc, n, sc->ws.depth > 0 ? sc->ws.depth : 1);
    if (rc != RT_OK) return rc;
    hipStream_t st = sc->ctx->stream;
    hipLaunchKernelGGL(k_aos_to_soa, grid_for(n), dim3(256), 0, st, rays_dev, n, sc->ws.pb.rays, sc->ws.cap);
    launch_trace(sc, false, nullptr, nullptr, n, sc->ws.pb.rays, t_min, t_max, prim_id_dev, nullptr, t_dev);
    HIP_OK(hipGetLastError());
    return RT_OK;
}

int rt_trace_closest_device_noalloc(rt_scene* sc, uint32_t n, const double* rays_dev, double t_min, double t_max, uint32_t* prim_id_dev, double* t_dev) {
    if (!sc || (n && (!rays_dev || !prim_id_dev || !t_dev))) return fail(RT_ERR_BAD_ARG, "rt_trace_closest_device_noalloc: null argument");
    if (n > RT_NOALLOC_MAX_RAYS) return fail(RT_ERR_BAD_ARG, "rt_trace_closest_device_noalloc: more than " + std::to_string(RT_NOALLOC_MAX_RAYS) + " rays per call (the scratch array is fixed)");
    if (n == 0) return RT_OK;
    HIP_OK(hipSetDevice(sc->ctx->device));
    hipStream_t st = sc->ctx->stream;
    unsigned char* base = sc->ctx->scratch;  // never the array's name: that is a host-side stand-in, not a device address
    unsigned long long* stats = (unsigned long long*)base;
    double* staged = (double*)(base + RT_NOALLOC_STATS_BYTES);
    HIP_OK(hipMemsetAsync(stats, 0, 16 * sizeof(unsigned long long), st));  // the array keeps the last call's contents; the ray blocks are rewritten below
    hipLaunchKernelGGL(k_aos_to_soa, grid_for(n), dim3(256), 0, st, rays_dev, n, staged, RT_NOALLOC_MAX_RAYS);
    launch_trace(sc, stats, RT_NOALLOC_MAX_RAYS, st, false, nullptr, nullptr, n, staged, t_min, t_max, prim_id_dev, nullptr, t_dev);
    HIP_OK(hipGetLastError());
    return RT_OK;
}

int rt_trace_closest(rt_scene* sc, uint32_t n, const double* rays, double t_min, double t_max, uint32_t* prim_id, uint32_t* inst_id, double* t,
                     uint64_t* counters) {
    if (!sc || (n && (!rays || !prim_id || !t))) return fail(RT_ERR_BAD_ARG, "rt_trace_closest: null argument");
    if (n == 0) { if (counters) for (int k = 0; k < 8; ++k) counters[k] = 0; return RT_OK; }
    RT_NOT_BUSY(sc, "rt_trace_closest");
    HIP_OK(hipSetDevice(sc->ctx->device));
    int rc = ws_ensure(sc, n, sc->ws.depth > 0 ? sc->ws.depth : 1);
    if (rc != RT_OK) return rc;
    Workspace& ws = sc->ws;
    hipStream_t st = sc->ctx->stream;
    std::vector<double> soa(RT_RAY_STRIDE * (size_t)n, 0.0);  // internal layout: one 64-byte line per ray
    for (uint32_t i = 0; i < n; ++i)
        for (int c = 0; c < 7; ++c) soa[(size_t)i * RT_RAY_STRIDE + c] = rays[(size_t)i * 7 + c];
    HIP_OK(hipMemcpyAsync(ws.pb.rays, soa.data(), soa.size() * sizeof(double), hipMemcpyHostToDevice, st));
    HIP_OK(hipMemsetAsync(ws.stats, 0, 16 * sizeof(unsigned long long), st));
    launch_trace(sc, counters != nullptr, nullptr, nullptr, n, ws.pb.rays, t_min, t_max, ws.pb.hit_prim, ws.pb.hit_inst, ws.pb.hit_t);
    HIP_OK(hipGetLastError());
    std::vector<uint32_t> hp(n), hi(n);
    unsigned long long st4[16];
    HIP_OK(hipMemcpyAsync(hp.data(), ws.pb.hit_prim, n * sizeof(uint32_t), hipMemcpyDeviceToHost, st));
    HIP_OK(hipMemcpyAsync(hi.data(), ws.pb.hit_inst, n * sizeof(uint32_t), hipMemcpyDeviceToHost, st));
    HIP_OK(hipMemcpyAsync(t, ws.pb.hit_t, n * sizeof(double), hipMemcpyDeviceToHost, st));
    HIP_OK(hipMemcpyAsync(st4, ws.stats, sizeof st4, hipMemcpyDeviceToHost, st));
    HIP_OK(hipStreamSynchronize(st));
    for (uint32_t i = 0; i < n; ++i) {
        if (hp[i] == RT_NONE) { prim_id[i] = RT_NONE; if (inst_id) inst_id[i] = RT_NONE; t[i] = 0.0; continue; }
        const rt_prim& p = sc->host_prims[hp[i]];
        prim_id[i] = p.object_id;
        if (inst_id) inst_id[i] = hi[i] != RT_NONE ? sc->host_prims[hi[i]].inst_id : p.inst_id;
    }
    if (counters) { counters[0] = st4[1]; counters[1] = st4[2]; counters[2] = st4[3]; counters[3] = st4[13]; counters[4] = st4[14]; counters[5] = counters[6] = counters[7] = 0; }
    return RT_OK;
}

int rt_primary_rays(rt_scene* sc, uint32_t width, uint32_t height, uint32_t sample, uint64_t seed, double* rays) {
    if (!sc || !rays || width < 2 || height < 2) return fail(RT_ERR_BAD_ARG, "rt_primary_rays: bad argument");
    const uint64_t n64 = (uint64_t)width * height;
    if (n64 > 0x7FFFFFFFull) return fail(RT_ERR_BAD_ARG, "rt_primary_rays: frame too large");
    const uint32_t n = (uint32_t)n64;
    HIP_OK(hipSetDevice(sc->ctx->device));
    int rc = ws_ensure(sc, n, sc->ws.depth > 0 ? sc->ws.depth : 1);
    if (rc != RT_OK) return rc;
    Workspace& ws = sc->ws;
    hipStream_t st = sc->ctx->stream;
    hipLaunchKernelGGL(k_raygen, grid_for(n), dim3(256), 0, st, sc->dev.camera, width, height, n, sample, n, (const uint32_t*)nullptr, (const uint32_t*)nullptr, seed,
                       ws.pb.rays, ws.cap, (PathInfo*)nullptr, 0u);
    double* aos = nullptr;
    HIP_OK(hipMalloc((void**)&aos, (size_t)n * 7 * sizeof(double)));
    hipLaunchKernelGGL(k_soa_to_aos, grid_for(n), dim3(256), 0, st, ws.pb.rays, n, ws.cap, aos);
    hipError_t e = hipMemcpyAsync(rays, aos, (size_t)n * 7 * sizeof(double), hipMemcpyDeviceToHost, st);
    if (e == hipSuccess) e = hipStreamSynchronize(st);
    (void)hipFree(aos);
    if (e != hipSuccess) return fail(RT_ERR_HIP, hipGetErrorString(e));
    return RT_OK;
}

// samples_per_pass == 0: a pure function of the parameters (see rt_render)
static uint32_t default_pass_size(uint32_t npix, int32_t max_depth) {
    const uint64_t by_paths = std::max<uint64_t>(1, (128ull << 20) / npix);
    const uint64_t by_mem = std::max<uint64_t>(1, (80ull << 30) / ((uint64_t)npix * ((uint64_t)max_depth * REC_STRIDE * 8 + 320)));
    return (uint32_t)std::min<uint64_t>(std::min(by_paths, by_mem), 256);
}

static int check_params(const rt_render_params* p) {
    if (!p) return fail(RT_ERR_BAD_ARG, "null params");
    if (p->width < 2 || p->height < 2) return fail(RT_ERR_BAD_ARG, "width/height must be >= 2 (u = (x+xi)/(width-1), main.rs:996)");
    if ((uint64_t)p->width * p->height > (1ull << 28)) return fail(RT_ERR_BAD_ARG, "frame too large");
    if (p->spp_end < p->spp_begin) return fail(RT_ERR_BAD_ARG, "spp_end < spp_begin");
    if (p->max_depth < 0 || p->max_depth > 4096) return fail(RT_ERR_BAD_ARG, "max_depth out of range");
    if (p->integrator < RT_INT_MIS || p->integrator > RT_INT_UNIFORM) return fail(RT_ERR_BAD_ARG, "unknown integrator");
    return RT_OK;
}

int rt_accum_zero(rt_context* ctx, float* accum_dev, uint32_t width, uint32_t height) {
    if (!ctx || !accum_dev) return fail(RT_ERR_BAD_ARG, "rt_accum_zero: null argument");
    if ((uint64_t)width * height > (1ull << 28)) return fail(RT_ERR_BAD_ARG, "rt_accum_zero: frame too large");
    HIP_OK(hipSetDevice(ctx->device));
    HIP_OK(hipMemsetAsync(accum_dev, 0, (size_t)width * height * 3 * sizeof(float), ctx->stream));
    return RT_OK;
}

int rt_render(rt_scene* sc, const rt_render_params* p, float* accum_dev, rt_stats* stats) {
    if (!sc || !accum_dev) return fail(RT_ERR_BAD_ARG, "rt_render: null argument");
    int rc = check_params(p);
    if (rc != RT_OK) return rc;
    RT_NOT_BUSY(sc, "rt_render");
    HIP_OK(hipSetDevice(sc->ctx->device));
    const uint32_t npix = p->width * p->height;
    const uint32_t spp = p->spp_end - p->spp_begin;
    if (stats) memset(stats, 0, sizeof *stats);
    if (spp == 0 || p->max_depth == 0) {  // depth <= 0: every sample returns 0 (integrator.rs:33-35)
        if (stats) stats->paths = (uint64_t)npix * spp;
        HIP_OK(hipStreamSynchronize(sc->ctx->stream));
        return RT_OK;
    }
    // Two passes are in flight at a time, each on its own stream with its own path buffers (a "lane"): every traversal launch
    // lasts as long as its slowest wave and the deep bounces carry few rays, so one pass alone leaves the machine part-idle
    // for ~1/8 of its time; the other lane's kernels fill those gaps.  A pass is as large as fits in a third of the free
    // HBM (at most 96 GiB) and 128M paths -- large passes matter for the same reason -- but not more than half the call's
    // samples, so that both lanes have work.  The counting kernel runs on one lane (same passes, same sums).
    uint32_t spass = p->samples_per_pass;
    if (spass == 0) {
        // A pure function of the parameters (never of free memory or of earlier calls): pass boundaries decide the order of
        // the fp32 additions into the accumulator, so resumed, chunked and sharded renders of the same samples must cut the
        // same passes to stay bit-identical.  Budget: 80 GiB of path state per lane (two lanes + scene fit the 288 GB), at most
        // 128M paths, at most half the call's samples so that both lanes have work.  If the buffers do not fit (a shared GPU),
        // the call fails with RT_ERR_OOM and the caller chooses samples_per_pass itself.
        spass = std::min(default_pass_size(npix, p->max_depth), (spp + 1u) / 2u);
    }
    spass = std::max(1u, std::min(spass, spp));
    if ((uint64_t)npix * spass > 0x7FFFFF00ull) return fail(RT_ERR_BAD_ARG, "samples_per_pass * pixels too large");
    const bool count_tests = (p->flags & RT_RENDER_COUNT_TESTS) != 0;
    const bool sort_rays = (p->flags & RT_RENDER_NO_SORT) == 0 && sc->sortable;
    const uint32_t n_pass = (spp + spass - 1u) / spass;
    int n_lanes = (n_pass >= 2 && !count_tests && (p->flags & RT_RENDER_ONE_LANE) == 0) ? 2 : 1;
    hipStream_t st = sc->ctx->stream;
    if (n_lanes == 2 && !sc->stream2) HIP_OK(hipStreamCreateWithFlags(&sc->stream2, hipStreamNonBlocking));
    rc = ws_ensure(sc, sc->ws, npix * spass, p->max_depth);
    if (rc != RT_OK) return rc;
    if (n_lanes == 2 && (uint64_t)npix * spass >= sc->ws2_failed_cap) n_lanes = 1;      // did not fit before
    if (n_lanes == 2 && ws_ensure(sc, sc->ws2, npix * spass, p->max_depth) != RT_OK) {  // no room for a second set of path buffers:
        (void)hipGetLastError();                                                         // same passes, one after the other
        sc->ws2_failed_cap = (uint64_t)npix * spass;
        n_lanes = 1;
    }

    struct LaneRun {
        Workspace* ws; hipStream_t st; EventTimer tm; uint64_t launches = 0, shadow_rays = 0; int rc = RT_OK; std::string err;
        LaneRun(Workspace* w, hipStream_t s, bool timed) : ws(w), st(s), tm(s, timed) {}
    };
    LaneRun lane[2] = {LaneRun(&sc->ws, st, stats != nullptr), LaneRun(&sc->ws2, sc->stream2, stats != nullptr)};
    // pass k adds its sums to the accumulator after pass k-1 did (fixed order: the frame does not depend on the lanes)
    std::vector<hipEvent_t> added(n_pass, nullptr);
    for (auto& e : added) HIP_OK(hipEventCreateWithFlags(&e, hipEventDisableTiming));
    std::mutex mu; std::condition_variable cv; uint32_t n_recorded = 0; bool broken = false;
    hipEvent_t e0, e1, e_lane2;
    HIP_OK(hipEventCreate(&e0)); HIP_OK(hipEventCreate(&e1)); HIP_OK(hipEventCreateWithFlags(&e_lane2, hipEventDisableTiming));
    HIP_OK(hipEventRecord(e0, st));
    if (n_lanes == 2) HIP_OK(hipStreamWaitEvent(sc->stream2, e0, 0));   // work queued on the caller's stream before this call comes first
    auto run_lane = [&](int li) -> int {
        LaneRun& L = lane[li];
        Workspace& ws = *L.ws;
        HIP_OK(hipSetDevice(sc->ctx->device));
        HIP_OK(hipMemsetAsync(ws.stats, 0, 16 * sizeof(unsigned long long), L.st));
        for (uint32_t k = (uint32_t)li; k < n_pass; k += (uint32_t)n_lanes) {
            const uint32_t s0 = p->spp_begin + k * spass;
            const uint32_t ns = std::min(spass, p->spp_end - s0);
            int r = run_paths(sc, ws, L.st, npix * ns, false, p->width, p->height, s0, p->max_depth, p->integrator, p->seed, count_tests, sort_rays, L.tm, L.launches, &L.shadow_rays, nullptr, (p->flags & RT_RENDER_LANE_WALK) != 0, false,
                              (p->flags & RT_RENDER_MATERIAL_BINS) != 0, (p->flags & (RT_RENDER_COUNT_STEPS | RT_RENDER_COUNT_TESTS)) != 0);
            if (r != RT_OK) return r;
            if (k > 0) {
                std::unique_lock<std::mutex> lk(mu);
                cv.wait(lk, [&] { return n_recorded >= k || broken; });
                if (broken) return fail(RT_ERR_HIP, "the other render lane failed");
                lk.unlock();
                HIP_OK(hipStreamWaitEvent(L.st, added[k - 1], 0));
            }
            L.tm.begin(CAT_OTHER);
            hipLaunchKernelGGL(k_accumulate, grid_for(npix), dim3(256), 0, L.st, ws.pb.radiance, ws.pb.info, ws.cap, npix, ns, accum_dev);
            L.tm.end();
            L.launches += 1;
            HIP_OK(hipEventRecord(added[k], L.st));
            { std::lock_guard<std::mutex> lk(mu); n_recorded = k + 1; }
            cv.notify_all();
        }
        return RT_OK;
    };
    auto guarded = [&](int li) {
        lane[li].rc = run_lane(li);
        if (lane[li].rc != RT_OK) {
            lane[li].err = g_err;
            { std::lock_guard<std::mutex> lk(mu); broken = true; }
            cv.notify_all();
        }
    };
    if (n_lanes == 2) {
        std::thread t2;
        try { t2 = std::thread(guarded, 1); } catch (...) { n_lanes = 1; }  // no second host thread: same passes on one lane (nothing has been launched yet)
        guarded(0);
        if (t2.joinable()) t2.join();
    } else guarded(0);
    unsigned long long st4[16] = {}, st4b[16] = {};
    hipError_t he = hipSuccess;
    if (n_lanes == 2) {
        he = hipEventRecord(e_lane2, sc->stream2);
        if (he == hipSuccess) he = hipStreamWaitEvent(st, e_lane2, 0);
        if (he == hipSuccess) he = hipMemcpyAsync(st4b, sc->ws2.stats, sizeof st4b, hipMemcpyDeviceToHost, sc->stream2);
    }
    if (he == hipSuccess) he = hipEventRecord(e1, st);
    if (he == hipSuccess) he = hipMemcpyAsync(st4, sc->ws.stats, sizeof st4, hipMemcpyDeviceToHost, st);
    if (n_lanes == 2) (void)hipStreamSynchronize(sc->stream2);
    hipError_t hs = hipStreamSynchronize(st);
    if (he == hipSuccess) he = hs;
    if (he == hipSuccess) he = hipGetLastError();
    for (int li = 0; li < n_lanes && rc == RT_OK; ++li)
        if (lane[li].rc != RT_OK) { rc = lane[li].rc; g_err = lane[li].err; }
    if (rc == RT_OK && he != hipSuccess) rc = fail(RT_ERR_HIP, std::string("rt_render: ") + hipGetErrorString(he));
    if (stats && rc == RT_OK) {
        float ms = 0.f;
        (void)hipEventElapsedTime(&ms, e0, e1);
        double by_cat[CAT_N] = {0, 0, 0, 0}; uint64_t n_by_cat[CAT_N] = {0, 0, 0, 0};
        for (int li = 0; li < n_lanes; ++li) lane[li].tm.collect(by_cat, CAT_N, n_by_cat);
        for (int k = 0; k < 16; ++k) st4[k] += st4b[k];
        stats->rays = st4[0]; stats->box_tests = st4[1]; stats->tri_tests = st4[2]; stats->sphere_tests = st4[3];
        stats->paths = (uint64_t)npix * spp; stats->launches = lane[0].launches + lane[1].launches;
        stats->ms_total = ms; stats->ms_trace = by_cat[CAT_TRACE] + by_cat[CAT_SHADOW]; stats->ms_shade = by_cat[CAT_SHADE]; stats->ms_other = by_cat[CAT_OTHER];
        stats->trace_launches = n_by_cat[CAT_TRACE] + n_by_cat[CAT_SHADOW]; stats->filter_violations = st4[13]; stats->trace_wave_steps = st4[15];
        stats->ms_trace_shadow = by_cat[CAT_SHADOW]; stats->trace_shadow_launches = n_by_cat[CAT_SHADOW];
        stats->shadow_rays = lane[0].shadow_rays + lane[1].shadow_rays; stats->samples_per_pass = spass; stats->lanes = (uint32_t)n_lanes;
        stats->walk_run_steps = st4[4]; stats->walk_tri_steps = st4[5]; stats->walk_slow_steps = st4[6]; stats->walk_iterations = st4[7];
    } else {
        double by_cat[CAT_N] = {0, 0, 0, 0}; uint64_t n_by_cat[CAT_N] = {0, 0, 0, 0};
        for (int li = 0; li < n_lanes; ++li) lane[li].tm.collect(by_cat, CAT_N, n_by_cat);   // releases the events
    }
    for (auto& e : added) (void)hipEventDestroy(e);
    (void)hipEventDestroy(e0); (void)hipEventDestroy(e1); (void)hipEventDestroy(e_lane2);
    return rc;
}

// One job = one rt_render call on ONE lane (own stream, own path buffers, own host thread for the per-bounce readbacks).
static void job_main(rt_job* J) {
    rt_scene* sc = J->sc;
    const rt_render_params& p = J->p;
    auto body = [&]() -> int {
        HIP_OK(hipSetDevice(sc->ctx->device));
        Workspace& ws = J->lane == 0 ? sc->ws : sc->ws2;
        hipStream_t st = J->lane == 0 ? sc->ctx->stream : sc->stream2;
        const uint32_t npix = p.width * p.height, spp = p.spp_end - p.spp_begin;
        EventTimer tm(st, true);
        uint64_t launches = 0, shadow_rays = 0;
        hipEvent_t e0, e1;
        HIP_OK(hipEventCreate(&e0)); HIP_OK(hipEventCreate(&e1));
        HIP_OK(hipEventRecord(e0, st));
        HIP_OK(hipMemsetAsync(ws.stats, 0, 16 * sizeof(unsigned long long), st));
        const bool sort_rays = (p.flags & RT_RENDER_NO_SORT) == 0 && sc->sortable;
        bool waited = J->seq == 0;
        for (uint32_t s0 = p.spp_begin; s0 < p.spp_end; s0 += J->spass) {
            const uint32_t ns = std::min(J->spass, p.spp_end - s0);
            int r = run_paths(sc, ws, st, npix * ns, false, p.width, p.height, s0, p.max_depth, p.integrator, p.seed, false, sort_rays, tm, launches, &shadow_rays, nullptr,
                              (p.flags & RT_RENDER_LANE_WALK) != 0, false, (p.flags & RT_RENDER_MATERIAL_BINS) != 0, (p.flags & (RT_RENDER_COUNT_STEPS | RT_RENDER_COUNT_TESTS)) != 0);
            if (r != RT_OK) return r;
            if (!waited) {   // the job begun before this one adds its sums first (fixed order of the fp32 additions)
                std::unique_lock<std::mutex> lk(sc->job_mu);
                sc->job_cv.wait(lk, [&] { return sc->jobs_recorded >= J->seq; });
                lk.unlock();
                HIP_OK(hipStreamWaitEvent(st, sc->job_done[(J->seq - 1) % 4], 0));
                waited = true;
            }
            tm.begin(CAT_OTHER);
            hipLaunchKernelGGL(k_accumulate, grid_for(npix), dim3(256), 0, st, ws.pb.radiance, ws.pb.info, ws.cap, npix, ns, J->accum);
            tm.end();
            launches += 1;
        }
        HIP_OK(hipEventRecord(sc->job_done[J->seq % 4], st));
        { std::lock_guard<std::mutex> lk(sc->job_mu); sc->jobs_recorded = J->seq + 1; }
        sc->job_cv.notify_all();
        HIP_OK(hipEventRecord(e1, st));
        unsigned long long st4[16] = {};
        HIP_OK(hipMemcpyAsync(st4, ws.stats, sizeof st4, hipMemcpyDeviceToHost, st));
        HIP_OK(hipStreamSynchronize(st));
        HIP_OK(hipGetLastError());
        float ms = 0.f;
        (void)hipEventElapsedTime(&ms, e0, e1);
        (void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
        double by_cat[CAT_N] = {0, 0, 0, 0}; uint64_t n_by_cat[CAT_N] = {0, 0, 0, 0};
        tm.collect(by_cat, CAT_N, n_by_cat);
        rt_stats& S = J->st;
        S.rays = st4[0]; S.paths = (uint64_t)npix * spp; S.launches = launches; S.ms_total = ms;
        S.ms_trace = by_cat[CAT_TRACE] + by_cat[CAT_SHADOW]; S.ms_shade = by_cat[CAT_SHADE]; S.ms_other = by_cat[CAT_OTHER];
        S.trace_launches = n_by_cat[CAT_TRACE] + n_by_cat[CAT_SHADOW]; S.trace_wave_steps = st4[15];
        S.ms_trace_shadow = by_cat[CAT_SHADOW]; S.trace_shadow_launches = n_by_cat[CAT_SHADOW]; S.shadow_rays = shadow_rays;
        S.samples_per_pass = J->spass; S.lanes = 1;
        S.walk_run_steps = st4[4]; S.walk_tri_steps = st4[5]; S.walk_slow_steps = st4[6]; S.walk_iterations = st4[7];
        return RT_OK;
    };
    J->rc = body();
    if (J->rc != RT_OK) {
        J->err = g_err;
        // the jobs behind this one must not wait for ever: count it as recorded (their sums are then meaningless, and their rt_render_end says why)
        { std::lock_guard<std::mutex> lk(sc->job_mu); sc->jobs_recorded = std::max(sc->jobs_recorded, J->seq + 1); }
        sc->job_cv.notify_all();
    }
}

static void lane_worker(rt_scene* sc, int lane) {
    for (;;) {
        rt_job* J = nullptr;
        {
            std::unique_lock<std::mutex> lk(sc->job_mu);
            sc->job_cv.wait(lk, [&] { return sc->workers_quit || (sc->lane_job[lane] && !sc->lane_job[lane]->started); });
            if (sc->workers_quit) return;
            J = sc->lane_job[lane];
            J->started = true;
        }
        job_main(J);
        { std::lock_guard<std::mutex> lk(sc->job_mu); J->finished = true; }
        sc->job_cv.notify_all();
    }
}

int rt_render_begin(rt_scene* sc, const rt_render_params* p, float* accum_dev, rt_job** out) {
    if (!sc || !accum_dev || !out) return fail(RT_ERR_BAD_ARG, "rt_render_begin: null argument");
    int rc = check_params(p);
    if (rc != RT_OK) return rc;
    if (p->flags & RT_RENDER_COUNT_TESTS) return fail(RT_ERR_BAD_ARG, "rt_render_begin: the counting kernel runs through rt_render");
    if (p->spp_end == p->spp_begin || p->max_depth == 0) return fail(RT_ERR_BAD_ARG, "rt_render_begin: empty sample range or depth 0 (nothing to render)");
    const int lane = (int)(sc->job_seq % 2);
    if (sc->lane_job[lane]) return fail(RT_ERR_BAD_ARG, "rt_render_begin: two jobs are in flight already (rt_render_end the older one first)");
    HIP_OK(hipSetDevice(sc->ctx->device));
    const uint32_t npix = p->width * p->height, spp = p->spp_end - p->spp_begin;
    uint32_t spass = p->samples_per_pass ? p->samples_per_pass : default_pass_size(npix, p->max_depth);   // one lane per job: not halved
    spass = std::max(1u, std::min(spass, spp));
    if ((uint64_t)npix * spass > 0x7FFFFF00ull) return fail(RT_ERR_BAD_ARG, "samples_per_pass * pixels too large");
    if (lane == 1 && !sc->stream2) HIP_OK(hipStreamCreateWithFlags(&sc->stream2, hipStreamNonBlocking));
    for (auto& e : sc->job_done) if (!e) HIP_OK(hipEventCreateWithFlags(&e, hipEventDisableTiming));
    rc = ws_ensure(sc, lane == 0 ? sc->ws : sc->ws2, npix * spass, p->max_depth);
    if (rc != RT_OK) return rc;
    rt_job* J = new (std::nothrow) rt_job();
    if (!J) return fail(RT_ERR_OOM, "rt_render_begin: host allocation failed");
    J->sc = sc; J->p = *p; J->accum = accum_dev; J->lane = lane; J->spass = spass;
    if (sc->job_seq == 0 || (!sc->lane_job[0] && !sc->lane_job[1])) {   // nothing in flight: this job has no predecessor to wait for
        std::lock_guard<std::mutex> lk(sc->job_mu);
        sc->jobs_recorded = sc->job_seq;
    }
    J->seq = sc->job_seq;
    if (lane == 1) {   // work queued on the caller's stream before this call comes first (lane 0 IS that stream)
        hipEvent_t e = nullptr;
        hipError_t he = hipEventCreateWithFlags(&e, hipEventDisableTiming);
        if (he == hipSuccess) he = hipEventRecord(e, sc->ctx->stream);
        if (he == hipSuccess) he = hipStreamWaitEvent(sc->stream2, e, 0);
        if (e) (void)hipEventDestroy(e);
        if (he != hipSuccess) { delete J; return fail(RT_ERR_HIP, hipGetErrorString(he)); }
    }
    if (!sc->worker[lane].joinable()) {
        try { sc->worker[lane] = std::thread(lane_worker, sc, lane); } catch (...) { delete J; return fail(RT_ERR_OOM, "rt_render_begin: cannot start a host thread"); }
    }
    { std::lock_guard<std::mutex> lk(sc->job_mu); sc->lane_job[lane] = J; }
    sc->job_cv.notify_all();
    sc->job_seq += 1;
    *out = J;
    return RT_OK;
}

int rt_render_end(rt_job* J, rt_stats* stats) {
    if (!J) return fail(RT_ERR_BAD_ARG, "rt_render_end: null job");
    rt_scene* sc = J->sc;
    if (sc->lane_job[J->lane] != J) return fail(RT_ERR_BAD_ARG, "rt_render_end: not a job in flight on its scene");
    {
        std::unique_lock<std::mutex> lk(sc->job_mu);
        sc->job_cv.wait(lk, [&] { return J->finished; });
        sc->lane_job[J->lane] = nullptr;
    }
    int rc = J->rc;
    if (rc != RT_OK) g_err = J->err;
    else if (stats) *stats = J->st;
    delete J;
    return rc;
}

int rt_capture_rays(rt_scene* sc, const rt_render_params* p, uint32_t bounce, int shadow, uint32_t stride, uint32_t max_n, uint32_t* n_out, uint32_t* queue_len,
                    double* rays, uint32_t* prim_id, uint32_t* inst_id, double* t, double* t_light) {
    if (!sc || !n_out || !rays || !prim_id || !inst_id || !t || !t_light || stride == 0) return fail(RT_ERR_BAD_ARG, "rt_capture_rays: bad argument");
    int rc = check_params(p);
    if (rc != RT_OK) return rc;
    if (p->spp_end != p->spp_begin + 1) return fail(RT_ERR_BAD_ARG, "rt_capture_rays: exactly one sample index");
    RT_NOT_BUSY(sc, "rt_capture_rays");
    if ((int32_t)bounce >= p->max_depth) return fail(RT_ERR_BAD_ARG, "rt_capture_rays: bounce >= max_depth");
    HIP_OK(hipSetDevice(sc->ctx->device));
    const uint32_t npix = p->width * p->height;
    rc = ws_ensure(sc, sc->ws, npix, p->max_depth);
    if (rc != RT_OK) return rc;
    Capture cap; cap.bounce = bounce; cap.shadow = shadow != 0; cap.stride = stride; cap.max_n = max_n;
    EventTimer tm(sc->ctx->stream, false);
    uint64_t launches = 0;
    HIP_OK(hipMemsetAsync(sc->ws.stats, 0, 16 * sizeof(unsigned long long), sc->ctx->stream));
    rc = run_paths(sc, sc->ws, sc->ctx->stream, npix, false, p->width, p->height, p->spp_begin, p->max_depth, p->integrator, p->seed, false,
                   (p->flags & RT_RENDER_NO_SORT) == 0 && sc->sortable, tm, launches, nullptr, &cap, (p->flags & RT_RENDER_LANE_WALK) != 0);
    HIP_OK(hipStreamSynchronize(sc->ctx->stream));
    if (rc != RT_OK) return rc;
    const uint32_t n = (uint32_t)cap.prim.size();
    *n_out = n; if (queue_len) *queue_len = cap.queue_len;
    memcpy(rays, cap.rays.data(), (size_t)n * 7 * sizeof(double));
    memcpy(t, cap.t.data(), (size_t)n * sizeof(double)); memcpy(t_light, cap.t_light.data(), (size_t)n * sizeof(double));
    for (uint32_t i = 0; i < n; ++i) {   // object ids, as rt_trace_closest reports them
        if (cap.prim[i] == RT_NONE) { prim_id[i] = RT_NONE; inst_id[i] = RT_NONE; continue; }
        const rt_prim& q = sc->host_prims[cap.prim[i]];
        prim_id[i] = q.object_id;
        inst_id[i] = cap.inst[i] != RT_NONE ? sc->host_prims[cap.inst[i]].inst_id : q.inst_id;
    }
    return RT_OK;
}

int rt_render_host(rt_scene* sc, const rt_render_params* p, float* accum_host, rt_stats* stats) {
    if (!sc || !accum_host) return fail(RT_ERR_BAD_ARG, "rt_render_host: null argument");
    int rc = check_params(p);
    if (rc != RT_OK) return rc;
    HIP_OK(hipSetDevice(sc->ctx->device));
    const size_t bytes = (size_t)p->width * p->height * 3 * sizeof(float);
    float* d = nullptr;
    HIP_OK(hipMalloc((void**)&d, bytes));
    hipError_t e = hipMemsetAsync(d, 0, bytes, sc->ctx->stream);
    if (e == hipSuccess) {
        rc = rt_render(sc, p, d, stats);
        if (rc == RT_OK) e = hipMemcpy(accum_host, d, bytes, hipMemcpyDeviceToHost);
    }
    (void)hipFree(d);
    if (rc != RT_OK) return rc;
    if (e != hipSuccess) return fail(RT_ERR_HIP, hipGetErrorString(e));
    return RT_OK;
}

int rt_resolve(rt_context* ctx, const float* accum_dev, uint32_t width, uint32_t height, uint32_t spp, int tonemap, float* out_dev) {
    if (!ctx || !accum_dev || !out_dev || spp == 0) return fail(RT_ERR_BAD_ARG, "rt_resolve: bad argument");
    if ((uint64_t)width * height > (1ull << 28)) return fail(RT_ERR_BAD_ARG, "rt_resolve: frame too large");
    HIP_OK(hipSetDevice(ctx->device));
    const uint32_t n3 = width * height * 3;
    hipLaunchKernelGGL(k_resolve, grid_for(n3), dim3(256), 0, ctx->stream, accum_dev, n3, 0.f, spp, tonemap, out_dev);
    HIP_OK(hipGetLastError());
    return RT_OK;
}

int rt_resolve_host(const float* accum_host, uint32_t width, uint32_t height, uint32_t spp, int tonemap, float* out_host) {
    if (!accum_host || !out_host || spp == 0) return fail(RT_ERR_BAD_ARG, "rt_resolve_host: bad argument");
    if ((uint64_t)width * height > (1ull << 28)) return fail(RT_ERR_BAD_ARG, "rt_resolve_host: frame too large");
    const size_t bytes = (size_t)width * height * 3 * sizeof(float);
    float *a = nullptr, *o = nullptr;
    HIP_OK(hipMalloc((void**)&a, bytes));
    hipError_t e = hipMalloc((void**)&o, bytes);
    if (e == hipSuccess) e = hipMemcpy(a, accum_host, bytes, hipMemcpyHostToDevice);
    if (e == hipSuccess) {
        const uint32_t n3 = width * height * 3;
        hipLaunchKernelGGL(k_resolve, grid_for(n3), dim3(256), 0, 0, a, n3, 0.f, spp, tonemap, o);
        e = hipMemcpy(out_host, o, bytes, hipMemcpyDeviceToHost);
    }
    (void)hipFree(a); (void)hipFree(o);
    if (e != hipSuccess) return fail(RT_ERR_HIP, hipGetErrorString(e));
    return RT_OK;
}

int rt_sample_radiance_sig(rt_scene* sc, uint32_t width, uint32_t height, uint32_t n, const uint32_t* pixel_index, const uint32_t* sample_index,
                           int32_t max_depth, int32_t integrator, uint64_t seed, double* out_rgb, uint64_t* out_sig);
int rt_sample_radiance(rt_scene* sc, uint32_t width, uint32_t height, uint32_t n, const uint32_t* pixel_index, const uint32_t* sample_index,
                       int32_t max_depth, int32_t integrator, uint64_t seed, double* out_rgb) {
    return rt_sample_radiance_sig(sc, width, height, n, pixel_index, sample_index, max_depth, integrator, seed, out_rgb, nullptr);
}
int rt_sample_radiance_sig(rt_scene* sc, uint32_t width, uint32_t height, uint32_t n, const uint32_t* pixel_index, const uint32_t* sample_index,
                           int32_t max_depth, int32_t integrator, uint64_t seed, double* out_rgb, uint64_t* out_sig) {
    if (!sc || (n && (!pixel_index || !sample_index || !out_rgb))) return fail(RT_ERR_BAD_ARG, "rt_sample_radiance: null argument");
    rt_render_params p{}; p.width = width; p.height = height; p.max_depth = max_depth; p.integrator = integrator;
    int rc = check_params(&p);
    if (rc != RT_OK) return rc;
    if (n == 0) return RT_OK;
    for (uint32_t i = 0; i < n; ++i)
        if (pixel_index[i] >= width * height) return fail(RT_ERR_BAD_ARG, "pixel index out of range");
    HIP_OK(hipSetDevice(sc->ctx->device));
    if (max_depth == 0) { memset(out_rgb, 0, (size_t)n * 3 * sizeof(double)); if (out_sig) memset(out_sig, 0, (size_t)n * sizeof(uint64_t)); return RT_OK; }
    RT_NOT_BUSY(sc, "rt_sample_radiance");
    rc = ws_ensure(sc, n, max_depth);
    if (rc != RT_OK) return rc;
    Workspace& ws = sc->ws;
    hipStream_t st = sc->ctx->stream;
    HIP_OK(hipMemcpyAsync(ws.in_pixel, pixel_index, n * sizeof(uint32_t), hipMemcpyHostToDevice, st));
    HIP_OK(hipMemcpyAsync(ws.in_sample, sample_index, n * sizeof(uint32_t), hipMemcpyHostToDevice, st));
    EventTimer tm(st, false);
    uint64_t launches = 0;
    if (out_sig && !ws.pb.sig) { rc = ws_alloc(ws, ws.pb.sig, ws.cap); if (rc != RT_OK) { ws.pb.sig = nullptr; return rc; } }
    rc = run_paths(sc, sc->ws, sc->ctx->stream, n, true, width, height, 0, max_depth, integrator, seed, false, sc->sortable, tm, launches, nullptr, nullptr, false, out_sig != nullptr);
    if (rc != RT_OK) return rc;
    if (out_sig) HIP_OK(hipMemcpyAsync(out_sig, ws.pb.sig, (size_t)n * sizeof(uint64_t), hipMemcpyDeviceToHost, st));
    std::vector<double> rad(3 * (size_t)ws.cap);
    HIP_OK(hipMemcpyAsync(rad.data(), ws.pb.radiance, rad.size() * sizeof(double), hipMemcpyDeviceToHost, st));
    HIP_OK(hipStreamSynchronize(st));
    for (uint32_t i = 0; i < n; ++i)
        for (int c = 0; c < 3; ++c) out_rgb[(size_t)i * 3 + c] = rad[(size_t)c * ws.cap + i];
    return RT_OK;
}

int rt_pdf_eval(rt_context* ctx, uint32_t n, uint32_t n_comp, const rt_pdf_desc* pdfs, const double* dirs, uint64_t seed, double* value, double* generated) {
    if (!ctx || (n && (!pdfs || !dirs || !value || !generated))) return fail(RT_ERR_BAD_ARG, "rt_pdf_eval: null argument");
    if (n_comp < 1 || n_comp > RT_PDF_MAX_COMP) return fail(RT_ERR_BAD_ARG, "rt_pdf_eval: n_comp must be 1..8");
    for (size_t k = 0; k < (size_t)n * n_comp; ++k)
        if (pdfs[k].kind > RT_PDF_CONE) return fail(RT_ERR_BAD_ARG, "rt_pdf_eval: unknown PDF kind");
    if (n == 0) return RT_OK;
    HIP_OK(hipSetDevice(ctx->device));
    const size_t b_desc = (size_t)n * n_comp * sizeof(rt_pdf_desc), b_dir = (size_t)n * 3 * sizeof(double), b_val = (size_t)n * sizeof(double);
    uint8_t* d = nullptr;
    HIP_OK(hipMalloc((void**)&d, b_desc + 2 * b_dir + b_val));
    rt_pdf_desc* d_desc = (rt_pdf_desc*)d; double* d_dir = (double*)(d + b_desc); double* d_gen = d_dir + (size_t)n * 3; double* d_val = d_gen + (size_t)n * 3;
    hipError_t e = hipMemcpy(d_desc, pdfs, b_desc, hipMemcpyHostToDevice);
    if (e == hipSuccess) e = hipMemcpy(d_dir, dirs, b_dir, hipMemcpyHostToDevice);
    if (e == hipSuccess) {
        hipLaunchKernelGGL(k_pdf_eval, grid_for(n), dim3(256), 0, ctx->stream, n, n_comp, d_desc, d_dir, seed, d_val, d_gen);
        e = hipStreamSynchronize(ctx->stream);
    }
    if (e == hipSuccess) e = hipMemcpy(value, d_val, b_val, hipMemcpyDeviceToHost);
    if (e == hipSuccess) e = hipMemcpy(generated, d_gen, b_dir, hipMemcpyDeviceToHost);
    (void)hipFree(d);
    if (e != hipSuccess) return fail(RT_ERR_HIP, hipGetErrorString(e));
    return RT_OK;
}

int rt_ieee_div_sqrt(rt_context* ctx, uint32_t n, const double* a, const double* b, double* q, double* s) {
    if (!ctx || (n && (!a || !b || !q || !s))) return fail(RT_ERR_BAD_ARG, "rt_ieee_div_sqrt: null argument");
    if (n == 0) return RT_OK;
    HIP_OK(hipSetDevice(ctx->device));
    double* d = nullptr;
    const size_t bytes = (size_t)n * sizeof(double);
    HIP_OK(hipMalloc((void**)&d, 4 * bytes));
    hipError_t e = hipMemcpy(d, a, bytes, hipMemcpyHostToDevice);
    if (e == hipSuccess) e = hipMemcpy(d + n, b, bytes, hipMemcpyHostToDevice);
    if (e == hipSuccess) {
        hipLaunchKernelGGL(k_div_sqrt, grid_for(n), dim3(256), 0, ctx->stream, n, d, d + n, d + 2 * (size_t)n, d + 3 * (size_t)n);
        e = hipStreamSynchronize(ctx->stream);
    }
    if (e == hipSuccess) e = hipMemcpy(q, d + 2 * (size_t)n, bytes, hipMemcpyDeviceToHost);
    if (e == hipSuccess) e = hipMemcpy(s, d + 3 * (size_t)n, bytes, hipMemcpyDeviceToHost);
    (void)hipFree(d);
    if (e != hipSuccess) return fail(RT_ERR_HIP, hipGetErrorString(e));
    return RT_OK;
}

}  // extern "C"
