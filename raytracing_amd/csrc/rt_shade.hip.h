// k_raygen, k_shade (+ material binning), k_unwind, k_accumulate, k_resolve and the small utility kernels.  Part of rt_device.hip: included there, nowhere else.
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
// COMPACT records (every integrator but MIS in a scene with lights: no light sample, no weight, no visibility test): only the first line of the table above can
// occur, so a record is the 32-byte block { brdf*cos (3), pdf } -- the product k_unwind would form first anyway, hence the same bits -- at a stride of 4 doubles
// in the same buffer.  k_shade / k_unwind are built in both forms; shading is half-way to the HBM roofline on the scenes without a BVH to walk, and a record
// was 80 of the 164 bytes it writes per ray.
enum { CREC_A_X = 0, CREC_A_Y, CREC_A_Z, CREC_PDF, CREC_STRIDE = 4 };

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

// Coherence key of a secondary ray.  A wave walks the UNION of its rays' records, so neighbours in the sorted queue should visit the same ones.
// What a ray visits is decided by two tests: the reference's weak per-axis box test -- a function of "which boxes lie ahead of the origin on each
// axis", i.e. of direction octant and ORIGIN -- and strict pruning, which is a function of the ray's LINE alone (a subtree is walked iff the line
// passes its box, rt_trace.hip.h): two rays on nearly the same line share their deep records however far apart their origins lie along it.  Key:
//     octant (3) | origin cell, Morton, RT_KEY_CELL bits per axis | direction bin, RT_KEY_DIR bits per axis of |d| / max|d| | foot point, 2 x RT_KEY_FOOT bits
// where the foot point is the line's point closest to the centre of the sort grid, taken in the two axes other than the dominant direction axis
// (which the direction bins imply) and Morton-interleaved.  Default 4 / 2 / 5 = 31 bits, four 8-bit radix passes.
// r03q: the diagnostic build's step log gives every sampled ray's exact visited set, tools/key_eval.py costs any order of those
// rays offline; it predicted 0.51 x the record steps of the first MIS shadow launch and 0.80 x of the first bounce for this layout against
// octant | cell 4 | direction 3 (24 bits, r01-r03p), and the GPU agreed: 2 289 -> 2 608 Mrays/s on the headline, frames bit-identical (a path's radiance
// does not depend on its place in a queue).  Measured beside it: 4/1/6 2 596, 4/1/5 2 565, 5/2/4 2 553, 3/2/6 2 545, 4/3/4 2 530, 23-bit layouts (three
// passes) 3/1/4 2 402 and 4/0/4 2 410.  Before the foot point: direction-major or finer-cell keys of 27-30 bits 2 114-2 301 (no gain), RT_KEY_LEGACY
// variants 1, 3, 4 no better than 2.
#ifndef RT_KEY_CELL
#define RT_KEY_CELL 4
#endif
#ifndef RT_KEY_DIR
#define RT_KEY_DIR 2
#endif
#ifndef RT_KEY_FOOT
#define RT_KEY_FOOT 5
#endif
RT_DEV uint32_t spread5(uint32_t v) { v &= 31u; v = (v | (v << 8)) & 0x100Fu; v = (v | (v << 4)) & 0x10C3u; v = (v | (v << 2)) & 0x1249u; return v; }   // bit k -> bit 3k
RT_DEV uint32_t spread10x2(uint32_t x) { x &= 0x3FFu; x = (x | (x << 8)) & 0x00FF00FFu; x = (x | (x << 4)) & 0x0F0F0F0Fu; x = (x | (x << 2)) & 0x33333333u; x = (x | (x << 1)) & 0x55555555u; return x; }   // bit k -> bit 2k
RT_DEV uint32_t sort_key(const SceneDev& sc, d3 o, d3 d) {
    const uint32_t oct = (d.x < 0.0 ? 1u : 0u) | (d.y < 0.0 ? 2u : 0u) | (d.z < 0.0 ? 4u : 0u);
    const float fx = ((float)o.x - sc.sort_min[0]) * sc.sort_scale[0], fy = ((float)o.y - sc.sort_min[1]) * sc.sort_scale[1], fz = ((float)o.z - sc.sort_min[2]) * sc.sort_scale[2];   // grid coordinates, 32 cells per axis
    const uint32_t cx = (uint32_t)fminf(fmaxf(fx, 0.f), 31.f), cy = (uint32_t)fminf(fmaxf(fy, 0.f), 31.f), cz = (uint32_t)fminf(fmaxf(fz, 0.f), 31.f);
    const float ax = fabsf((float)d.x), ay = fabsf((float)d.y), az = fabsf((float)d.z);
    const float amax = fmaxf(fmaxf(ax, ay), fmaxf(az, 1e-30f));
#ifdef RT_KEY_LEGACY   /* the 24-bit key of r01-r03p, for A/B: octant | cell 4 | direction 3 */
    const float inv8 = 7.999f / amax;
    const uint32_t d9 = ((uint32_t)(ax * inv8) & 7u) | (((uint32_t)(ay * inv8) & 7u) << 3) | (((uint32_t)(az * inv8) & 7u) << 6);
    const uint32_t m4 = spread5(cx >> 1) | (spread5(cy >> 1) << 1) | (spread5(cz >> 1) << 2);
    return (oct << 21) | ((m4 & 0xFFFu) << 9) | d9;
#else
    const uint32_t cq = 5u - RT_KEY_CELL;
    const uint32_t cell = spread5(cx >> cq) | (spread5(cy >> cq) << 1) | (spread5(cz >> cq) << 2);
    const float invd = ((float)(1u << RT_KEY_DIR) - 0.001f) / amax;
    const uint32_t dm = (1u << RT_KEY_DIR) - 1u;
    const uint32_t dbin = RT_KEY_DIR ? (((uint32_t)(ax * invd) & dm) | (((uint32_t)(ay * invd) & dm) << RT_KEY_DIR) | (((uint32_t)(az * invd) & dm) << (2 * RT_KEY_DIR))) : 0u;
    // foot point: r = o - centre in world units per axis, minus its component along the unit direction, back to grid units (0.5 = the centre)
    const float dx = (float)d.x, dy = (float)d.y, dz = (float)d.z;
    const float il = 1.0f / fmaxf(__builtin_sqrtf(dx * dx + dy * dy + dz * dz), 1e-30f);
    const float nx = dx * il, ny = dy * il, nz = dz * il;
    const float rx = ((float)o.x - sc.sort_min[0]) - 16.0f / sc.sort_scale[0], ry = ((float)o.y - sc.sort_min[1]) - 16.0f / sc.sort_scale[1], rz = ((float)o.z - sc.sort_min[2]) - 16.0f / sc.sort_scale[2];
    const float along = rx * nx + ry * ny + rz * nz;
    const float gx = (rx - along * nx) * sc.sort_scale[0] * (1.0f / 32.0f) + 0.5f, gy = (ry - along * ny) * sc.sort_scale[1] * (1.0f / 32.0f) + 0.5f, gz = (rz - along * nz) * sc.sort_scale[2] * (1.0f / 32.0f) + 0.5f;
    const uint32_t dom = (ax >= ay && ax >= az) ? 0u : (ay >= az ? 1u : 2u);
    const float fu = dom == 0u ? gy : gx, fv = dom == 2u ? gy : gz;
    const float ftop = (float)((1u << RT_KEY_FOOT) - 1u);
    const uint32_t qu = (uint32_t)fminf(fmaxf(fu * (float)(1u << RT_KEY_FOOT), 0.f), ftop), qv = (uint32_t)fminf(fmaxf(fv * (float)(1u << RT_KEY_FOOT), 0.f), ftop);
    const uint32_t foot = spread10x2(qu) | (spread10x2(qv) << 1);
    return ((((oct << (3 * RT_KEY_CELL)) | cell) << (3 * RT_KEY_DIR)) | dbin) << (2 * RT_KEY_FOOT) | foot;
#endif
}
#ifdef RT_KEY_LEGACY
#define RT_SORT_BITS 24u
#else
#define RT_SORT_BITS (3u + 3u * RT_KEY_CELL + 3u * RT_KEY_DIR + 2u * RT_KEY_FOOT)
#endif
static_assert(RT_SORT_BITS <= 32u && RT_KEY_CELL <= 5 && RT_KEY_FOOT <= 10, "the sort key is one 32-bit word");

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
// INTEG: the integrator as a compile-time constant (-1: the run-time argument).  The body is one function for four integrators; built per integrator the
// dead arms go (the light sample and its 30-odd live doubles exist in the MIS build only) and the allocator has that much less to keep.
template <bool COMPACT, int INTEG>
RT_DEV void shade_block(const SceneDev& sc, const PathBuffers& pb, const uint32_t* __restrict__ queue, const uint32_t* __restrict__ order, const uint32_t* __restrict__ n_ptr,
                        uint32_t bounce, int32_t max_depth, int32_t integrator_arg, uint64_t seed,
                        uint32_t* __restrict__ next_queue, uint32_t* __restrict__ next_keys, uint32_t* __restrict__ next_count,
                        uint32_t* __restrict__ shadow_queue, uint32_t* __restrict__ shadow_keys, uint32_t* __restrict__ shadow_count) {
    const int32_t integrator = INTEG >= 0 ? INTEG : integrator_arg;
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
        uint32_t n_records = bounce;   // records this path holds when it ends here: one per level before this one, plus this level's if it ends by depth
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
                    if (COMPACT) {
                        const d3 a = brdf * cos_theta;   // k_unwind: ((brdf * cos) * L) / pdf
                        double4* rec = (double4*)(pb.records + ((size_t)bounce * cap + path) * CREC_STRIDE);
                        *rec = make_double4(a.x, a.y, a.z, pdf);
                    } else {
                        double* rec = pb.records + ((size_t)bounce * cap + path) * REC_STRIDE;
                        rec[REC_BRDF_X] = brdf.x; rec[REC_BRDF_Y] = brdf.y; rec[REC_BRDF_Z] = brdf.z;
                        rec[REC_COS] = cos_theta; rec[REC_PDF] = pdf; rec[REC_W] = w;
                        rec[REC_CL_X] = c_l.x; rec[REC_CL_Y] = c_l.y; rec[REC_CL_Z] = c_l.z;
                        rec[REC_T_LIGHT] = t_light;
                    }
                    if ((int32_t)(bounce + 1u) >= max_depth) {  // next level has depth <= 0: returns 0 without tracing (integrator.rs:33-35)
                        terminal = mk(0.0, 0.0, 0.0);
                        done = true;
                        n_records = bounce + 1u;
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
        // a path ends exactly once: its terminal radiance and its record count go out together (one 32-byte block; the count used to be a
        // scattered 4-byte write per shaded ray into the path's info block)
        if (done) { double4* tb = (double4*)(pb.terminal + (size_t)path * 4u); *tb = make_double4(terminal.x, terminal.y, terminal.z, (double)n_records); }
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
template <bool COMPACT, int INTEG = -1> __global__ __launch_bounds__(RT_SHADE_BLOCK) RT_SHADE_OCCUPANCY void k_shade(RT_SHADE_PARAMS) { shade_block<COMPACT, INTEG>(RT_SHADE_ARGS); }
#ifndef RT_SHADE_DENSE_WAVES
#define RT_SHADE_DENSE_WAVES 4
#endif
template <bool COMPACT, int INTEG = -1> __global__ __launch_bounds__(RT_SHADE_BLOCK) __attribute__((amdgpu_waves_per_eu(RT_SHADE_DENSE_WAVES, RT_SHADE_DENSE_WAVES))) void k_shade_dense(RT_SHADE_PARAMS) { shade_block<COMPACT, INTEG>(RT_SHADE_ARGS); }

// ------------------------------------------------------------------------------------------------
// k_unwind / k_accumulate / k_resolve
// ------------------------------------------------------------------------------------------------
template <bool COMPACT>
__global__ __launch_bounds__(256) void k_unwind(PathBuffers pb, uint32_t n) {
    const uint32_t p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= n) return;
    const size_t cap = pb.cap;
    const double4 tb = *(const double4*)(pb.terminal + (size_t)p * 4u);
    d3 L = mk(tb.x, tb.y, tb.z);
    unsigned long long sg = 0ull;
    for (int k = (int)tb.w - 1; k >= 0; --k) {
        d3 e = mk(0.0, 0.0, 0.0);  // emitted of a scattering material
        if (COMPACT) {
            const double4 r = *(const double4*)(pb.records + ((size_t)k * cap + p) * CREC_STRIDE);
            L = e + (mk(r.x, r.y, r.z) * L) / r.w;
            continue;
        }
        const double* rec = pb.records + ((size_t)k * cap + p) * REC_STRIDE;
        d3 brdf = mk(rec[REC_BRDF_X], rec[REC_BRDF_Y], rec[REC_BRDF_Z]);
        double cos_theta = rec[REC_COS], pdf = rec[REC_PDF], t_light = rec[REC_T_LIGHT];
        bool visible = false;
        if (t_light == t_light) {  // a light sample was taken: visible iff the shadow ray hit and |t_scene - t_light| <= 1e-4 (integrator.rs:82-86)
            double t_shadow = rec[REC_T_SHADOW];
            visible = (t_shadow == t_shadow) && !(fabs(t_shadow - t_light) > 0.0001);
            sg += sig_hash((uint32_t)k, SIG_VIS, visible ? 1ull : 0ull);
        }
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
// run_paths' counts ring: [2b + 1] = shadow rays of bounce b.  One wave adds their sum to a statistics slot (rt_stats.shadow_rays).
__global__ __launch_bounds__(64) void k_sum_shadow_counts(const uint32_t* __restrict__ counts, uint32_t depth, unsigned long long* __restrict__ out) {
    uint64_t s = 0;
    for (uint32_t b = threadIdx.x; b < depth; b += 64u) s += counts[2u * b + 1u];
    s = wave_sum_u64(s);
    if (threadIdx.x == 0 && s) atomicAdd(out, (unsigned long long)s);
}
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
