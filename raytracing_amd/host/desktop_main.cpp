// rt_desktop: counterpart of the reference's `desktop` binary (desktop/src/main.rs:923-1028): pick a
// scene, an integrator, spp and depth; render on the GPU through the C ABI; ACES; write output.ppm.
// The reference edits these settings in source (main.rs:924-953); here they are flags.
#include "../../include/rt_abi.h"
#include "desktop.hpp"
#include <chrono>
#include <cstdlib>

using namespace renderer;

static const char* arg(int argc, char** argv, const char* key, const char* dflt) {
    for (int i = 1; i + 1 < argc; ++i) if (!strcmp(argv[i], key)) return argv[i + 1];
    return dflt;
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
        rt_stats st{};
        auto t0 = std::chrono::steady_clock::now();                              // main.rs:976
        CHECK(rt_render_host(gs, &p, acc.data(), &st));
        auto t1 = std::chrono::steady_clock::now();
        printf("Rendering complete in %.3fs (%s: %.2f Mrays/s, %.3g paths/s)\n", std::chrono::duration<double>(t1 - t0).count(), integrator->name(),
               st.rays / st.ms_total / 1e3, st.paths / st.ms_total * 1e3);
        CHECK(rt_resolve_host(acc.data(), width, height, samples_per_pixel, 1, px.data()));  // main.rs:1009-1011
        Image img = Image::new_({width, height});
        for (size_t i = 0; i < img.data.size(); ++i) img.put((uint32_t)(i % width), (uint32_t)(i / width), DVec3(px[3 * i], px[3 * i + 1], px[3 * i + 2]));
        printf("Writing PPM file %s\n", out.c_str());
        desktop::write_image(img, out);                                          // main.rs:1027
        rt_scene_destroy(gs); rt_context_destroy(ctx);
    } catch (const std::exception& e) { fprintf(stderr, "error: %s\n", e.what()); return 1; }
    return 0;
}
