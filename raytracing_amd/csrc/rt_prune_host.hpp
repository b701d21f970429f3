// Host side of strict pruning: which transform space is pruned, its error constants, the direction grid of the certificate (cert_raster) and the
// per-cluster plane lists.  Part of rt_device.hip: included there, nowhere else.
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
    std::vector<uint8_t> node_ok;          // per node of the scene-level range (index - root_node): every primitive below can be covered (empty: pruning off)
    std::vector<double> node_box;          // 6 per such node: f64 box of the primitives below in space xf (valid where node_ok)
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
    R.node_ok = ok; R.node_box = box;   // the production stream's re-bracketed hierarchy derives the local boxes of its own inner nodes from the leaves'
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

// Per INSTANCE primitive: what inst_line_misses (rt_trace.hip.h) needs to skip the whole sub-array.  An instance qualifies when every
// primitive of its sub-array is a triangle in the sub-array's own space (prepare_scene has validated: no nested transform) with finite,
// moderate coordinates and a shape factor of at least RT_PRUNE_MIN_SHAPE, and there are at most RT_INST_CULL_MAX_TRIS of them.
// Constants as in build_prune: tau_coef = 2^-21 / shape_min, delta_coef = 2 tau_coef, radius = max |v0|_inf + max edge 1-norm.
static void build_inst_cull(const rt_blob_header& h, const rt_node* nodes, const rt_prim* prims, const rt_geom* geom, std::vector<InstCull>& cull, std::vector<InstPlane>& planes) {
    cull.assign((size_t)h.n_prims, InstCull{});
    for (uint32_t p = 0; p < h.n_prims; ++p) {
        if (prims[p].kind != RT_PRIM_INSTANCE) continue;
        InstCull c{};
        double lo[3] = {INFINITY, INFINITY, INFINITY}, hi[3] = {-INFINITY, -INFINITY, -INFINITY};
        double shape_min = 1.0, e_max = 0.0, v_max = 0.0;
        bool good = true;
        std::vector<uint32_t> tris;
        for (uint32_t i = prims[p].index; i < prims[p].sub_end && good; ++i)
            for (uint32_t q : {nodes[i].prim0, nodes[i].prim1}) {
                if (q == RT_NONE) continue;
                if (geom[q].kind != RT_PRIM_TRIANGLE || geom[q].xform != RT_NONE) { good = false; break; }
                if (std::find(tris.begin(), tris.end(), q) == tris.end()) tris.push_back(q);   // a 1-object leaf lists its primitive once; be safe about repeats
                if (tris.size() > RT_INST_CULL_MAX_TRIS) { good = false; break; }
            }
        good = good && !tris.empty();
        const uint32_t first_plane = (uint32_t)planes.size();
        for (size_t t = 0; t < tris.size() && good; ++t) {
            const double* v0 = geom[tris[t]].d; const double* e1 = v0 + 3; const double* e2 = v0 + 6;
            for (int k = 0; k < 9; ++k) good = good && std::isfinite(v0[k]);
            const double E1 = std::fabs(e1[0]) + std::fabs(e1[1]) + std::fabs(e1[2]), E2 = std::fabs(e2[0]) + std::fabs(e2[1]) + std::fabs(e2[2]);
            const double V = std::max(std::max(std::fabs(v0[0]), std::fabs(v0[1])), std::fabs(v0[2]));
            good = good && E1 >= 0x1p-100 && E1 <= 0x1p100 && E2 >= 0x1p-100 && E2 <= 0x1p100 && V <= 0x1p100;
            if (!good) break;
            const double n[3] = {e1[1] * e2[2] - e1[2] * e2[1], e1[2] * e2[0] - e1[0] * e2[2], e1[0] * e2[1] - e1[1] * e2[0]};
            const double len = std::sqrt(n[0] * n[0] + n[1] * n[1] + n[2] * n[2]);
            const double shape = len / (E1 * E2);
            good = good && shape >= RT_PRUNE_MIN_SHAPE * (1.0 + 0x1p-20);
            if (!good) break;
            shape_min = std::min(shape_min, shape); e_max = std::max(e_max, std::max(E1, E2)); v_max = std::max(v_max, V);
            InstPlane pl;
            for (int k = 0; k < 3; ++k) pl.n[k] = n[k] / len;
            pl.q = v0[0] * pl.n[0] + v0[1] * pl.n[1] + v0[2] * pl.n[2];
            planes.push_back(pl);
            for (int k = 0; k < 3; ++k) {   // vertices v0, v0 + e1, v0 + e2 (sums rounded outward)
                const double a = v0[k] + e1[k], b = v0[k] + e2[k];
                lo[k] = std::min(lo[k], std::min(v0[k], std::min(std::nextafter(a, -INFINITY), std::nextafter(b, -INFINITY))));
                hi[k] = std::max(hi[k], std::max(v0[k], std::max(std::nextafter(a, INFINITY), std::nextafter(b, INFINITY))));
            }
        }
        if (!good) { planes.resize(first_plane); continue; }
        for (int k = 0; k < 3; ++k) { c.lo[k] = lo[k]; c.hi[k] = hi[k]; }
        c.tau_coef = 0x1p-21 / shape_min; c.delta_coef = 0x1p-20 / shape_min; c.radius = v_max + e_max;
        c.ok = 1u; c.n_planes = (uint32_t)(planes.size() - first_plane); c.first_plane = first_plane;
        cull[p] = c;
    }
}
