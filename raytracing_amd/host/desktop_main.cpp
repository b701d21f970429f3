// rt_desktop: counterpart of the reference's `desktop` binary (desktop/src/main.rs:923-1028): pick a
// scene, an integrator, spp and depth; render on the GPU through the C ABI; ACES; write output.ppm.
// The reference edits these settings in source (main.rs:924-953); here they are flags.
// Progressive mode (--chunk N [--checkpoint FILE] [--resume]): the frame is rendered N samples at a time into fp32 sums that are
// persisted after every chunk (include/rt_checkpoint.h).  ctrl-c finishes the chunk in flight, writes the partial image and exits 0
// -- the reference's handler (main.rs:970-974) dumps whatever tiles are done -- and --resume continues at the recorded sample.
#include "../../include/rt_abi.h"
#include "../../include/rt_checkpoint.h"
#include "desktop.hpp"
#include <chrono>
#include <csignal>
#include <cstdlib>
#include <unistd.h>

using namespace renderer;

static const char* arg(int argc, char** argv, const char* key, const char* dflt) {
    for (int i = 1; i + 1 < argc; ++i) if (!strcmp(argv[i], key)) return argv[i + 1];
    return dflt;
}
static volatile std::sig_atomic_t g_interrupted = 0;
static void on_sigint(int) { g_interrupted = 1; }

static uint32_t crc32_ieee(const uint8_t* p, size_t n) {  // zlib's CRC-32 (what raytracing_amd/progressive.py computes with zlib.crc32)
    static uint32_t table[256];
    if (!table[1]) for (uint32_t i = 0; i < 256; ++i) { uint32_t c = i; for (int k = 0; k < 8; ++k) c = (c & 1u) ? 0xEDB88320u ^ (c >> 1) : c >> 1; table[i] = c; }
    uint32_t c = 0xFFFFFFFFu;
    for (size_t i = 0; i < n; ++i) c = table[(c ^ p[i]) & 0xFFu] ^ (c >> 8);
    return c ^ 0xFFFFFFFFu;
}
static bool save_checkpoint(const std::string& path, const rt_checkpoint_header& h, const std::vector<float>& acc) {
    const std::string tmp = path + ".tmp";
    FILE* f = fopen(tmp.c_str(), "wb");
    if (!f) return false;
    bool ok = fwrite(&h, sizeof h, 1, f) == 1 && fwrite(acc.data(), sizeof(float), acc.size(), f) == acc.size();
    ok = (fflush(f) == 0) && ok;
    ok = (fsync(fileno(f)) == 0) && ok;                     // on disk before the rename makes it the checkpoint
    ok = (fclose(f) == 0) && ok;
    return ok && rename(tmp.c_str(), path.c_str()) == 0;   // a reader sees the old file or the new one
}
static bool load_checkpoint(const std::string& path, uint32_t width, uint32_t height, rt_checkpoint_header& h, std::vector<float>& acc, std::string& why) {
    FILE* f = fopen(path.c_str(), "rb");
    if (!f) { why = "cannot open " + path; return false; }
    bool ok = fread(&h, sizeof h, 1, f) == 1 && !memcmp(h.magic, RT_CHECKPOINT_MAGIC, 8);
    if (!ok) { why = "not a render checkpoint"; fclose(f); return false; }
    if (h.width != width || h.height != height) { why = "the checkpoint holds another frame size"; fclose(f); return false; }  // before anything is sized from the file
    acc.assign((size_t)h.width * h.height * 3, 0.f);
    ok = fread(acc.data(), sizeof(float), acc.size(), f) == acc.size() && fgetc(f) == EOF;
    fclose(f);
    if (!ok) { why = "checkpoint size does not match its header"; return false; }
    if (!(h.spp_begin <= h.spp_done && h.spp_done <= h.spp_end)) { why = "checkpoint sample counters out of order"; return false; }
    return true;
}

#define CHECK(call) do { int rc_ = (call); if (rc_ != RT_OK) { fprintf(stderr, "%s failed (%d): %s\n", #call, rc_, rt_last_error()); return 1; } } while (0)

int main(int argc, char** argv) {
    const uint32_t width = (uint32_t)atoi(arg(argc, argv, "--width", "1920"));   // main.rs:924
    const double aspect_ratio = 16.0 / 9.0;
    const uint32_t height = (uint32_t)((double)width / aspect_ratio);            // main.rs:926
    const std::string scene_name = arg(argc, argv, "--scene", "random");         // main.rs:931
    const uint32_t samples_per_pixel = (uint32_t)atoi(arg(argc, argv, "--spp", "10"));  // main.rs:943
    const int max_depth = atoi(arg(argc, argv, "--depth", "5"));                 // main.rs:944
    const std::string integ = arg(argc, argv, "--integrator", "mis");            // main.rs:950-953
    const std::string out = arg(argc, argv, "--out", "output.ppm");
    const std::string assets = arg(argc, argv, "--assets", "assets");
    bvh_axis_seed() = strtoull(arg(argc, argv, "--bvh-seed", "0"), nullptr, 10);
    {   // --bvh-axis seeded|min_area: which of the trees bvh.rs:53 can draw gets built (renderer.hpp "Split-axis source")
        const std::string ax = arg(argc, argv, "--bvh-axis", "seeded");
        if (ax != "seeded" && ax != "min_area") { fprintf(stderr, "rt_desktop: --bvh-axis must be seeded or min_area\n"); return 2; }
        bvh_axis_mode() = ax == "seeded" ? BVH_AXIS_SEEDED : BVH_AXIS_MIN_AREA;
    }
    MultipleImportanceSampleIntegrator mis; ImportanceSampleLightIntegrator lis; BRDFSampledPathIntegrator brdf; UniformSampledPathIntegrator uni;
    const Integrator* integrator = integ == "mis" ? (Integrator*)&mis : integ == "light_is" ? (Integrator*)&lis : integ == "brdf" ? (Integrator*)&brdf : (Integrator*)&uni;
    try {
        Scene scene = desktop::scene_by_name(scene_name, desktop::Assets{assets}, false);
        printf("Rendering scene with %u samples per pixel, %d max bounces, at a resolution of %ux%u\n", samples_per_pixel, max_depth, width, height);
        std::vector<uint8_t> blob = scene.flatten();
        rt_context* ctx = nullptr; rt_scene* gs = nullptr;
        CHECK(rt_context_create(atoi(arg(argc, argv, "--device", "0")), nullptr, &ctx));
        CHECK(rt_scene_upload(ctx, blob.data(), blob.size(), &gs));
        rt_render_params p{};
        p.width = width; p.height = height; p.spp_begin = 0; p.spp_end = samples_per_pixel; p.max_depth = max_depth;
        p.integrator = integrator->abi_id(); p.seed = strtoull(arg(argc, argv, "--seed", "24301"), nullptr, 10);
        std::vector<float> acc((size_t)width * height * 3), px(acc.size());
        const uint32_t chunk = (uint32_t)atoi(arg(argc, argv, "--chunk", "0"));
        const std::string ckpt = arg(argc, argv, "--checkpoint", "");
        const uint32_t stop_after = (uint32_t)atoi(arg(argc, argv, "--stop-after", "0"));   // test hook: as if ctrl-c arrived once this many samples are in
        bool resume = false;
        for (int i = 1; i < argc; ++i) resume = resume || !strcmp(argv[i], "--resume");
        if ((resume || !ckpt.empty()) && chunk == 0) { fprintf(stderr, "error: --checkpoint / --resume need --chunk N (samples per checkpoint)\n"); return 1; }
        uint32_t spp_done = 0;
        rt_stats st{};
        auto t0 = std::chrono::steady_clock::now();                              // main.rs:976
        if (chunk == 0) {
            CHECK(rt_render_host(gs, &p, acc.data(), &st));
            spp_done = samples_per_pixel;
        } else {
            rt_checkpoint_header h{};
            memcpy(h.magic, RT_CHECKPOINT_MAGIC, 8);
            h.width = width; h.height = height; h.spp_begin = 0; h.spp_done = 0; h.spp_end = samples_per_pixel; h.max_depth = max_depth;
            h.integrator = p.integrator; h.rank = 0; h.world = 1; h.seed = p.seed;
            h.chunk = chunk;   // chunk boundaries fix the order of the fp32 additions: a resume must cut the same chunks
            h.scene_id = ((uint64_t)(blob.size() & 0xFFFFFFFFu) << 32) | crc32_ieee(blob.data(), blob.size());
            if (resume) {
                rt_checkpoint_header f{}; std::string why;
                if (ckpt.empty() || !load_checkpoint(ckpt, width, height, f, acc, why)) { fprintf(stderr, "error: --resume: %s\n", ckpt.empty() ? "no --checkpoint given" : why.c_str()); return 1; }
                if (f.width != h.width || f.height != h.height || f.max_depth != h.max_depth || f.integrator != h.integrator || f.seed != h.seed ||
                    f.scene_id != h.scene_id || f.spp_begin != 0 || f.spp_done > h.spp_end || (f.chunk != 0 && f.chunk != h.chunk)) {
                    fprintf(stderr, "error: --resume: the checkpoint was written for another scene or other settings (frame, depth, integrator, seed, chunk)\n"); return 1;
                }
                h.spp_done = f.spp_done;
                printf("Resuming at sample %u of %u\n", h.spp_done, h.spp_end);
            }
            std::signal(SIGINT, on_sigint);                                      // main.rs:970
            std::vector<float> part(acc.size());
            while (h.spp_done < h.spp_end) {
                // chunk boundaries sit at multiples of `chunk` whatever the resume point: same sums, same order as an uninterrupted run
                p.spp_begin = h.spp_done; p.spp_end = std::min((h.spp_done / chunk + 1) * chunk, h.spp_end);
                rt_stats cs{};
                CHECK(rt_render_host(gs, &p, part.data(), &cs));
                for (size_t i = 0; i < acc.size(); ++i) acc[i] += part[i];
                st.rays += cs.rays; st.paths += cs.paths; st.ms_total += cs.ms_total;
                h.spp_done = p.spp_end;
                if (!ckpt.empty() && !save_checkpoint(ckpt, h, acc)) { fprintf(stderr, "error: cannot write checkpoint %s\n", ckpt.c_str()); return 1; }
                if (g_interrupted || (stop_after && h.spp_done >= stop_after)) break;
            }
            spp_done = h.spp_done;
            if (spp_done < samples_per_pixel) printf("Interrupted after %u of %u samples per pixel\n", spp_done, samples_per_pixel);
        }
        auto t1 = std::chrono::steady_clock::now();
        printf("Rendering complete in %.3fs (%s: %.2f Mrays/s, %.3g paths/s)\n", std::chrono::duration<double>(t1 - t0).count(), integrator->name(),
               st.rays / std::max(st.ms_total, 1e-9) / 1e3, st.paths / std::max(st.ms_total, 1e-9) * 1e3);
        if (spp_done == 0) { fprintf(stderr, "nothing rendered\n"); return 1; }
        CHECK(rt_resolve_host(acc.data(), width, height, spp_done, 1, px.data()));  // main.rs:1009-1011 (a partial frame divides by the samples it holds)
        Image img = Image::new_({width, height});
        for (size_t i = 0; i < img.data.size(); ++i) img.put((uint32_t)(i % width), (uint32_t)(i / width), DVec3(px[3 * i], px[3 * i + 1], px[3 * i + 2]));
        printf("Writing PPM file %s\n", out.c_str());
        auto ends_with = [&](const char* e) { const size_t n = strlen(e); return out.size() >= n && out.compare(out.size() - n, n, e) == 0; };
        if (ends_with(".pfm")) desktop::write_pfm(img, out);
        else if (ends_with(".png")) desktop::write_png(img, out);
        else desktop::write_image(img, out);                                     // main.rs:1027
        rt_scene_destroy(gs); rt_context_destroy(ctx);
    } catch (const std::exception& e) { fprintf(stderr, "error: %s\n", e.what()); return 1; }
    return 0;
}
