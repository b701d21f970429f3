// k_walk_setup / k_walk: the per-lane walk of the trace stream (opt-in A/B counterpart of k_trace).  Part of rt_device.hip: included there, nowhere else.
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
