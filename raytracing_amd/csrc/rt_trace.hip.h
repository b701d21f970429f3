// k_trace and everything it is made of: the trace stream's record layout, the fp32 filters with their error bounds, strict pruning and
// the certificate (device side), the wave-coherent traversal.  Part of rt_device.hip (one translation unit): included there, nowhere else.
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
    uint32_t dlog_wave;   // wave index if this wave's steps are logged (diag_log), else 0xFFFFFFFF
#endif
};
#ifdef RT_TRACE_DIAG
// Step log of sampled waves (diagnostic build, RT_DIAG_LOG=<dir>): one entry per record step = { wave, stream offset, mask of the lanes that take
// part }.  A ray's visited records do not depend on its wave-mates, so the log gives every sampled ray's exact visited set and any regrouping
// of those rays can be costed offline (tools/lane_log.py): the union of the visited offsets of a group is the step count of a wave made of it.
__device__ uint4* g_dlog; __device__ unsigned int g_dlog_n, g_dlog_cap, g_dlog_every, g_dlog_group;
__device__ float* g_dray; __device__ unsigned int g_dray_n, g_dray_cap;   // the sampled waves' rays: 8 floats each = { bits of wave * 64 + lane, o.xyz, d.xyz, t_light or 0 } (tools/key_eval.py)
RT_DEV void diag_log(uint32_t wave, uint32_t off, uint64_t m) {
    if (wave != 0xFFFFFFFFu && __builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u)) == 0u) {
        const unsigned int i = atomicAdd(&g_dlog_n, 1u);
        if (i < g_dlog_cap) g_dlog[i] = make_uint4(wave, off, (uint32_t)m, (uint32_t)(m >> 32));
    }
}
#endif

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
RT_DEV lanemask strict_filter(const StrictFilter& f, const Window& w, float dis) {   // dis: PruneState::dis
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
        "v_max_f32 %[f0], %[f0], %[dis]\n\t"       // a lane that may not prune: far = +inf, never "near > far"
        "v_cmp_gt_f32_e64 %[m], %[n0], %[f0]"
        : [n0] "=&v"(n0), [n1] "=&v"(n1), [n2] "=&v"(n2), [f0] "=&v"(f0), [f1] "=&v"(f1), [f2] "=&v"(f2v), [m] "=&s"(m)
        : [nxa] "v"(nx.x), [nxb] "v"(nx.y), [nya] "v"(ny.x), [nyb] "v"(ny.y), [nza] "v"(nz.x), [nzb] "v"(nz.y),
          [fxa] "v"(fx.x), [fxb] "v"(fx.y), [fya] "v"(fy.x), [fyb] "v"(fy.y), [fza] "v"(fz.x), [fzb] "v"(fz.y), [dis] "v"(dis));
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
    uint32_t any_suspect;  // wave-uniform, one SGPR (a bool becomes a 64-bit lane mask)
    float dis;         // per lane: -inf if the lane may prune (a bit of `enable`), +inf if not.  The strict filter takes max(far, dis) before its last compare, so a
                       // lane that may not prune never reports "near > far": one VALU instruction instead of `& enable`, whose SGPR pair the allocator kept
                       // spilled (two v_readlane per inner step): 2 238 -> 2 277 Mrays/s, frames bit-identical (profiles/r03m_ab_spills.txt)
    uint32_t none;     // wave-uniform: 1 iff enable == 0 (kept as an integer: the hot loop tests it together with the record kind)
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

// Strict pruning at the level of a whole instance (cornell: six Transformed(BVHNode::from_mesh(cube)) objects, main.rs:161-188).  The
// reference enters the sub-BVH whenever its weak per-axis test passes the sub-array's root box -- for a ray inside the room that is
// most of the six -- and then tests boxes and triangles that cannot be hit.  By Claim 2 a triangle the reference
// ACCEPTS has the ray's line within tau of it, unless the ray is nearly parallel to the triangle's plane and starts within delta of
// that plane.  So when (o, d) -- the ray in the instance's object space, the bits the reference computes (transform.rs:25-32) -- is not
// suspect for any triangle of the instance and its LINE misses the box of those triangles inflated by tau, the sub-walk accepts
// nothing: it leaves the running best, every later decision and the final hit unchanged, and is skipped.  Everything is f64 here (the
// slack 2^-40 relative covers the roundings of this test itself).  The counting kernel walks anyway and counts accepts after a "miss".
RT_DEV bool inst_line_misses(const SceneDev& sc, uint32_t p, d3 o, d3 d, d3 inv) {   // inv = 1 / d, the divisions the sub-walk needs anyway (bounding_box.rs:29)
    if (sc.inst_cull == nullptr) return false;
    cptr<InstCull> ic = as_const(sc.inst_cull) + p;
    if (ic->ok == 0u) return false;
    const double ad[3] = {fabs(d.x), fabs(d.y), fabs(d.z)}, oo[3] = {o.x, o.y, o.z}, ii[3] = {inv.x, inv.y, inv.z};
    const double D = fmax(fmax(ad[0], ad[1]), ad[2]), O = fmax(fmax(fabs(o.x), fabs(o.y)), fabs(o.z));
    if (!(D >= 0x1p-100 && D <= 0x1p100 && O <= 0x1p100)) return false;   // also NaN
    const double tau = ic->tau_coef * (O + ic->radius) * (1.0 + 0x1p-20), delta = ic->delta_coef * (O + ic->radius), thr = RT_CERT_EPS * D;
    double tn = -__builtin_inf(), tf = __builtin_inf();
    bool miss = false;
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        const double lo = ic->lo[k] - tau, hi = ic->hi[k] + tau;
        if (ad[k] >= 0x1p-200 && fabs(ii[k]) <= 0x1p300) {   // 1/d finite and non-zero: slab parameters by multiplication (errors far inside the slack)
            const double t0 = (lo - oo[k]) * ii[k], t1 = (hi - oo[k]) * ii[k];
            const double s = 0x1p-40 * ((fabs(lo) + fabs(hi)) + fabs(oo[k])) * fabs(ii[k]);
            tn = fmax(tn, fmin(t0, t1) - s); tf = fmin(tf, fmax(t0, t1) + s);
        } else if (ad[k] < 0x1p-200) miss = miss || oo[k] < lo || oo[k] > hi;   // parallel to the slab and outside it: the line never enters
    }
    miss = miss || tn > tf;
    if (!miss) return false;
    cptr<InstPlane> pl = as_const(sc.inst_planes) + ic->first_plane;
    const uint32_t np = ic->n_planes;
    for (uint32_t i = 0; i < np; ++i) {   // suspect for a triangle: nearly parallel to its plane AND within delta of it (then the reference may accept garbage)
        const double nx = pl[i].n[0], ny = pl[i].n[1], nz = pl[i].n[2];
        if (fabs(d.x * nx + d.y * ny + d.z * nz) < thr && fabs((o.x * nx + o.y * ny + o.z * nz) - pl[i].q) <= delta) return false;
    }
    return true;
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
#ifdef RT_ORACLE_BEST
        if (sc.oracle_best == 1u) return false;
#endif
        const d3 xinv = mk(1.0 / xd.x, 1.0 / xd.y, 1.0 / xd.z);
        const bool culled = pass && inst_line_misses(sc, p, xo, xd, xinv);
        // BVHNode::hit starts with its own box (bvh.rs:92-94): decided here, in the reference's arithmetic, before the sub-walk sets anything up --
        // after a first hit most instances fail it (their slab lies beyond the running best), and setting a sub-walk up costs as much as walking it
        bool root_ok = true;
        if (!COUNT) {
            cptr<rt_node> sub_nodes = as_const(sc.nodes);
            const RT_CONST rt_node& rn = sub_nodes[(uint32_t)g.d[0]];
            const double best = h.best;
            const bool plain_ray = ray_is_plain(xo, xinv) && t_min <= best && sc.boxes_finite != 0u;
            root_ok = (rn.flags & RT_NODE_NO_BOX) != 0u || !plain_ray ||   // NaN-capable rays: the sub-walk decides with the literal form
                      (box_axis_fast(rn.bmin[0], rn.bmax[0], xo.x, xinv.x, t_min, best) & box_axis_fast(rn.bmin[1], rn.bmax[1], xo.y, xinv.y, t_min, best) &
                       box_axis_fast(rn.bmin[2], rn.bmax[2], xo.z, xinv.z, t_min, best));
        }
        bool enter = COUNT ? pass : (pass && !culled && root_ok);
#ifdef RT_ORACLE_BEST   /* experiment build: where do an instance's instructions go?  1: no instance is entered; 2: everything but the sub-walk */
        if (sc.oracle_best == 2u) enter = false;
#endif
        if (__builtin_amdgcn_ballot_w64(enter) != 0ull) {   // wave-uniform: nobody left -> the sub-array is not walked at all
            RayInRegisters sub; sub.r.o = xo; sub.r.d = xd; sub.r.time = ray.time; sub.rinv = xinv;
            const double before = h.best;
            traverse<1, COUNT, INST, RayInRegisters, HIT>(sc, (uint32_t)g.d[0], sub, h, enter, t_min, p, cnt);  // first node of the sub-array
            changed = (double)h.best != before;
            if (COUNT) cnt.bad += (changed && culled) ? 1u : 0u;   // an accept inside an instance the cull would have skipped
        }
        RT_RECONVERGE();
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
// `not_strict` (wave-uniform): 0 iff the record is an inner one, whose local box may prune (StrictFilter).
template <int B, int MODE, bool COUNT, class RAYSRC, class HIT>
RT_DEV lanemask box_step(const Window& w, const BoxFilter& bf, const PruneState& ps, uint32_t not_strict, uint32_t off, uint32_t& wait, uint32_t rec,
                         cptr<rt_node> nodes, const RAYSRC& src, HIT& h, double t_min, LaneCount& cnt) {
    const lanemask m_act = mask_reached(off, wait);
#ifdef RT_TRACE_DIAG
    cnt.lane_steps += mask_lane(m_act) ? 1u : 0u;
#endif
    if (COUNT) cnt.box += (mask_lane(m_act) && !(rec & RT_NODE_NO_BOX)) ? 1u : 0u;
    lanemask m_pass = 0ull;
    if (MODE == BOX_PLAIN) {
        lanemask cp = 0ull, cf = 0ull, sp = 0ull;
        // (a subtree that is never pruned carries the local box (-inf, +inf): near = -inf, far = +inf for every enabled lane -- their fp32 1/d is finite
        // and non-zero -- so the filter itself answers "not pruned"; testing for it here would cost every step two scalar instructions)
        uint32_t no_strict_test = not_strict | ps.none;
        if (!(__builtin_constant_p(not_strict) && not_strict != 0u)) {   // (the leaf-first call passes the constant 1: nothing to test, nothing to hide from the optimiser)
            no_strict_test = (uint32_t)__builtin_amdgcn_readfirstlane((int)no_strict_test);
            asm("" : "+s"(no_strict_test));
        }
        if (no_strict_test == 0u) {
            sp = strict_filter<B + 8>(ps.sf, w, ps.dis);
            if (ps.any_suspect) {  // a subtree that holds one of the ray's suspect leaves is not pruned
                const uint32_t span = wword<B + 6>(w) - off;
                sp &= ~(mask_lt_u32(ps.s0 - off, span) | mask_lt_u32(ps.s1 - off, span) | mask_lt_u32(ps.s2 - off, span));
            }
        }
        const lanemask sp_apply = COUNT ? 0ull : sp;  // the counting kernel walks the reference's own visited set
        // RT_STRICT_FIRST: when every ray that is here is pruned the reference's own test decides nothing -- m_pass stays 0 (ONE exit: an
        // early return leaves the compiler a "returned early" flag to carry as a lane mask through the rest of the step)
        if (COUNT || !RT_STRICT_FIRST || (m_act & ~sp_apply) != 0ull) {
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

// a wave-uniform value pinned to an SGPR: every value the cursor is assigned goes through this, so that the loop-carried cursor stays scalar
// (left to itself the compiler homes it in a VGPR: a v_mov per assignment, a v_cndmask per select and a v_readfirstlane per step)
RT_DEV uint32_t scalar_u32(uint32_t x) { x = (uint32_t)__builtin_amdgcn_readfirstlane((int)x); asm("" : "+s"(x)); return x; }

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
                off = scalar_u32(nxt_ > off ? nxt_ : off);                                                   \
            }                                                                                      \
        }                                                                                          \
        RT_RECONVERGE();                                                                           \
    }
    const uint32_t* node_off = full ? sc.node_off_full : sc.node_off;
    first = LEVEL == 0 ? (full ? sc.root_off_full : sc.root_off) : as_const(node_off)[first];
    uint32_t wait = active ? first : 0xFFFFFFFFu;
    // strict pruning: top-level walks of NaN-free rays in scenes that have a pruned space; the walk starts in that space
    const bool prune = LEVEL == 0 && MODE == BOX_PLAIN && sc.prune_on != 0u;
    uint32_t off = scalar_u32(first), cur_xf = prune ? sc.prune_xf : RT_NONE;
    lanemask clean = 0ull;
    PruneState ps;
    ps.enable = 0ull; ps.none = 1u; ps.s0 = ps.s1 = ps.s2 = 0xFFFFFFFFu; ps.any_suspect = 0; ps.dis = __builtin_inff();
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
        ps.any_suspect = (uint32_t)__builtin_amdgcn_readfirstlane(__builtin_amdgcn_ballot_w64(c && sus.leaf[0] != 0xFFFFFFFFu) != 0ull ? 1 : 0);
        asm("" : "+s"(ps.any_suspect));
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
        ps.enable = 0ull; ps.none = 1u;
        {
            const RayS ray = src.get();
            d3 xo, xd;
            apply_xform<MODE == BOX_PLAIN>(sc, cur_xf, ray, xo, xd);
            tri_filter_init(tf, xo, xd);
            if (prune && cur_xf == sc.prune_xf) {  // inner records carry boxes of THIS space only
                const bool ok = strict_filter_init(ps.sf, xo, xd, sc);
                ps.enable = clean & __builtin_amdgcn_ballot_w64(ok); ps.none = (uint32_t)__builtin_amdgcn_readfirstlane(ps.enable == 0ull ? 1 : 0);
                ps.dis = mask_lane(ps.enable) ? -__builtin_inff() : __builtin_inff();
#ifdef RT_TRACE_DIAG
                cnt.diag_flags |= (active && !ok) ? 4u : 0u;
#endif
            } else {
                ps.sf.inv_xy = ps.sf.invz_cnx = ps.sf.cn_yz = ps.sf.cp_xy = ps.sf.cpz_ = (f2){0.f, 0.f};
                ps.dis = __builtin_inff();
            }
        }
        uint32_t want = cur_xf;
        for (;;) {
            Window w;
#ifdef RT_TRACE_DIAG
            const uint64_t c0 = __builtin_readcyclecounter();
#endif
            off = scalar_u32(off);
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
            if (LEVEL == 0) diag_log(cnt.dlog_wave, off | ((rec & RT_STREAM_KIND) >> 2), mask_reached(off, wait));   // records are 16-byte aligned: the kind rides in the low bits
#endif
            // "a TRI2 leaf in the space the rays are in" as ONE integer test (made opaque to the optimiser, which would otherwise turn it back into
            // two booleans and combine those as 64-bit lane masks: 10 scalar instructions instead of 5 on every step)
            uint32_t not_leaf_here = ((rec & RT_STREAM_KIND) ^ RT_STREAM_TRI2) | (wword<8>(w) ^ cur_xf);
            not_leaf_here = (uint32_t)__builtin_amdgcn_readfirstlane((int)not_leaf_here);   // wave-uniform by construction; a no-op where the compiler knows it
            asm("" : "+s"(not_leaf_here));
            if (RT_LEAF_FIRST && MODE == BOX_PLAIN && !COUNT && not_leaf_here == 0u) {
                // Common leaf, rays already in its space: triangle filter FIRST.  The leaf's box only gates its own
                // triangle tests (skip == next record either way), so it is evaluated for a wave only when some ray is
                // not certainly rejected by both triangles -- the weak box test of the reference passes most leaves
                // of a ray that has no hit yet, so the box would rarely have saved the triangle filter anyway.
                const lanemask m_act = mask_reached(off, wait);
                lanemask rejA, rejB;
                tri_filter(tf, w, rejA, rejB);
                bool changed = false;
                if ((m_act & ~(rejA & rejB)) != 0ull) {
                    const lanemask m_pass = box_step<0, MODE, COUNT>(w, bf, ps, 1u, off, wait, rec, nodes, src, h, t_min, cnt);
                    if ((m_pass & ~(rejA & rejB)) != 0ull) changed = tri2_exact<COUNT, MODE>(sc, node, off, m_pass, rejA, rejB, src, cur_xf, h, t_min, inst_prim, bf, cnt);
                    RT_RECONVERGE();
                }
                off = scalar_u32(off + RT_STREAM_REC_TRI2);
                RT_SHADOW_FINISH(changed)
                continue;
            }
            lanemask m_pass = box_step<0, MODE, COUNT>(w, bf, ps, rec & RT_STREAM_KIND, off, wait, rec, nodes, src, h, t_min, cnt);
            if (m_pass == 0ull) {
                off = scalar_u32(skip);
            } else if ((rec & RT_STREAM_KIND) == RT_STREAM_INNER) {
                off = scalar_u32(off + RT_STREAM_REC_INNER);
                // the window already holds the next record's header: if that is an inner node too, take it in the same step
                // (no load, no wait, no trip round the loop) -- a quarter of all steps on the pagoda before strict pruning
                const uint32_t rec2 = wword<23>(w);
                if ((rec2 & RT_STREAM_KIND) == RT_STREAM_INNER) {
#ifdef RT_TRACE_DIAG
                    cnt.steps += 1;
                    if (LEVEL == 0) diag_log(cnt.dlog_wave, off | ((rec & RT_STREAM_KIND) >> 2), mask_reached(off, wait));   // records are 16-byte aligned: the kind rides in the low bits
#endif
                    cnt.wsteps += 1u;
                    m_pass = box_step<16, MODE, COUNT>(w, bf, ps, 0u, off, wait, rec2, nodes, src, h, t_min, cnt);
                    off = scalar_u32(m_pass == 0ull ? wword<22>(w) : off + RT_STREAM_REC_INNER);
                }
            } else if ((rec & RT_STREAM_KIND) == RT_STREAM_SPH2) {  // one or two spheres: exact f64 test (shape.rs:93-121) on inline geometry
                // the spheres' own transform is applied on the spot: only the triangle filter needs the walk to BE in a space
                off = scalar_u32(off + RT_STREAM_REC_SPH2);
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
                off = scalar_u32(off + RT_STREAM_REC_TRI2);
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
                off = scalar_u32(off + RT_STREAM_REC_LEAF);
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

#define RT_LANES_AUTO 0x80000000u   /* k_trace's `lanes` argument: bit 31 | least lanes (7 bits) << 24 | target waves (24 bits) */
#ifndef RT_TRACE_BLOCK
#define RT_TRACE_BLOCK 64    /* threads per k_trace block: ONE wave.  A block keeps its LDS until its last wave ends, and the walks are long-tailed (mean 260 steps,
                                * max 5 600): with four waves per block the finished waves' LDS stayed allocated and kept new blocks out.  256 / 128 / 64: 1 960 / 2 018 / 2 090 Mrays/s (r02k) */
#endif
#ifndef RT_TRACE_WAVES
#define RT_TRACE_WAVES 6  /* waves per SIMD the register allocator must leave room for */
#endif
#ifndef RT_TRACE_WAVES_INST
#define RT_TRACE_WAVES_INST 4   /* scenes with instances (cornell): the nested sub-walk doubles the live state (168 VGPRs left alone); measured at 4K MIS: 2 / 3 / 4 / 5 / 6 waves = 1 711 / 2 111 / 2 165 / 1 516 / 1 458 Mrays/s */
#endif
#ifndef RT_TRACE_WAVES_REGS
#define RT_TRACE_WAVES_REGS 4   /* 128 VGPRs + 88 B of scratch; 3 / 4 / 5 waves: earth 6 455 / 6 717 / 6 818, random spheres 2 188 / 2 296 / 2 095, broken cube 6 900 / 6 983 / 6 867 Mrays/s */
#endif
// rays: one 64-byte block per path (ox,oy,oz,dx | dy,dz,time,pad); outputs indexed by path.
// REGS: keep the f64 ray in registers instead of LDS -- for scenes whose leaves are mostly not triangles, where
// (almost) every leaf step runs an exact f64 test and would read the ray back from LDS each time.
// SHADOW: the rays are MIS visibility rays whose block carries t_light in its last slot; see RayFromLdsT.
template <bool COUNT, bool INST, bool REGS, bool SHADOW = false>
__global__ __launch_bounds__(RT_TRACE_BLOCK) __attribute__((amdgpu_waves_per_eu(INST ? RT_TRACE_WAVES_INST : REGS ? RT_TRACE_WAVES_REGS : RT_TRACE_WAVES))) void k_trace(SceneDev sc, const uint32_t* __restrict__ queue, const uint32_t* __restrict__ n_ptr, uint32_t n_fixed,
                                               const double* __restrict__ rays, uint32_t cap, double t_min, double t_max,
                                               uint32_t* __restrict__ out_prim, uint32_t* __restrict__ out_inst, double* __restrict__ out_t,
                                               unsigned long long* __restrict__ stats, uint32_t lanes, uint32_t t_stride, PathHit* __restrict__ out_hit,
                                               const uint32_t* __restrict__ remap) {
    // `lanes` rays per wave (64, or fewer for incoherent queues: the wave walks the UNION of its rays' visited records)
    // remap (k_walk's left-overs): the launch covers the n queue SLOTS listed in remap instead of slots 0..n-1
    const uint32_t n = n_ptr ? *n_ptr : n_fixed;
    if (lanes & RT_LANES_AUTO) {   // small queues: rays per wave from the TRUE count (the host only knows a bound): the largest power of two <= 64 that still
        const uint32_t target = lanes & 0xFFFFFFu, least = (lanes >> 24) & 0x7Fu;   // leaves about `target` waves, at least `least`
        uint32_t l = 64u;
        while (l > least && n <= target * (l >> 1)) l >>= 1;
        lanes = l;
    }
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
#ifdef RT_ORACLE_BEST   /* experiment: what would KNOWING the hit distance be worth?  The walk starts from the t a previous launch found */
    if (sc.oracle_best != 0u && out_hit && valid) { const double b = out_hit[slot].t; if (b == b) h.best = b * (1.0 + 0x1p-30) + 0x1p-30; }
#endif
    LaneCount cnt; cnt.box = cnt.tri = cnt.sph = cnt.bad = 0; cnt.pruned_until = 0; cnt.uncertified = 0; cnt.wsteps = 0;
#ifdef RT_TRACE_DIAG
    cnt.steps = cnt.leaf_steps = cnt.cyc_wait = cnt.lane_steps = cnt.exact_box = cnt.exact_tri = 0; cnt.cyc_xbox = cnt.cyc_xtri = cnt.cyc_setup = cnt.cyc_leaf = cnt.leaf_open = 0; cnt.cyc_total = __builtin_readcyclecounter(); cnt.diag_flags = 0;
    cnt.dlog_wave = (g_dlog != nullptr && !COUNT && ((tid >> 6) / g_dlog_group % g_dlog_every) == 0u) ? (tid >> 6) : 0xFFFFFFFFu;   // groups of consecutive waves
    if (cnt.dlog_wave != 0xFFFFFFFFu && g_dray != nullptr && valid) {
        const unsigned int i = atomicAdd(&g_dray_n, 1u);
        if (i < g_dray_cap) {
            const RayS r = src.get();
            float* q = g_dray + 8u * (size_t)i;
            q[0] = __builtin_bit_cast(float, tid); q[1] = (float)r.o.x; q[2] = (float)r.o.y; q[3] = (float)r.o.z; q[4] = (float)r.d.x; q[5] = (float)r.d.y; q[6] = (float)r.d.z;
            q[7] = SHADOW ? (float)src.tl() : 0.f;
        }
    }
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
