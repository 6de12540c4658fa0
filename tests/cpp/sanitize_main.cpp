// Sanitizer driver (CPU build only: -fsanitize=address,undefined): runs the host-side code of the product (DDS/BC6H decode,
// scene -> ModelBounds extraction, host constant math) and the CPU oracle (test infrastructure) over the shipped fixtures,
// random inputs and hostile inputs (truncated files, garbage blocks, empty and ragged sizes). Any report from ASan/UBSan
// ends the process with a non-zero status. SURVEY.md section 5, "Race detection / sanitizers": the reference has only the D3D12
// debug layer (Source/RHI/DX12Device.cpp:82-91).
//
//   g++ -std=c++17 -O1 -g -fsanitize=address,undefined -fno-sanitize-recover=undefined -ffp-contract=off -I include \
//       tests/cpp/sanitize_main.cpp csrc/dds.cpp csrc/scene.cpp csrc/host_math.cpp oracle/ur_oracle.cpp -pthread
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <random>
#include <sstream>
#include <string>
#include <vector>

#include "ur_assets.h"
#include "ur_host.h"
#include "ur_hotpath.h"
#include "ur_scene.h"

extern "C" {
uint32_t uro_hzb_layout(uint32_t w, uint32_t h, ur_mip_desc* mips, uint32_t* mip_count);
void uro_build_hzb(const float* depth, uint32_t src_w, uint32_t src_h, float* hzb, const ur_mip_desc* mips, uint32_t MipCount);
void uro_cull_indirect_args(const uint32_t* constants, const ur_float4* bounds, const float* hzb, const ur_mip_desc* mips, void* indirect_args,
                            uint32_t* stats2, uint32_t* visible_idx, uint32_t* visible_count, uint32_t index_base);
void uro_deferred_lighting(const ur_scene_constants* scene, const ur_half4* gbuf_a, const ur_half4* gbuf_b, const uint32_t* gbuf_c,
                           const float* shadow, const ur_half4* env_cube, uint32_t env_base, uint32_t env_mips, const uint16_t* lut, uint32_t lut_w,
                           uint32_t lut_h, ur_half4* hdr, uint32_t w, uint32_t h, uint32_t row0, uint32_t rows, uint8_t* fragile);
void uro_sky_atmosphere(const ur_sky_constants* sky, const float* depth, ur_half4* hdr, uint32_t w, uint32_t h, uint32_t row0, uint32_t rows);
size_t uro_env_cube_texels(uint32_t base, uint32_t mips);
void uro_stage_env_cube(const ur_half4* src, uint32_t base, uint32_t mips, ur_half4* dst);
void uro_tonemap(const ur_tonemap_constants* K, const ur_half4* hdr, const float* exposure_ev, uint32_t* out, uint32_t count);
void uro_temporal_aa(const ur_half4* current, const ur_half4* history, ur_half4* output, float HistoryWeight, uint32_t UseHistory, uint32_t W,
                     uint32_t H, uint32_t row0, uint32_t rows);
void uro_set_threads(int n);
uint16_t uro_f2h(float f);
}

static int g_fail = 0;
#define CHECK(c) do { if (!(c)) { std::printf("FAIL %s:%d %s\n", __FILE__, __LINE__, #c); ++g_fail; } } while (0)

static std::string slurp(const std::string& path)
{
    std::ifstream f(path, std::ios::binary);
    std::stringstream ss;
    ss << f.rdbuf();
    return ss.str();
}

static void test_dds(const std::string& assets)
{
    const std::string cube = slurp(assets + "/output_pmrem.dds"), lut = slurp(assets + "/PreintegratedGF.dds");
    CHECK(!cube.empty() && !lut.empty());
    ur_dds_info info{};
    CHECK(ur_dds_parse(cube.data(), cube.size(), &info) == UR_ASSET_OK);
    CHECK(info.is_cube && info.width == 256 && info.mip_count == 9 && info.dxgi_format == 96);
    // exact-size heap buffers: an out-of-bounds texel write is an ASan report
    std::vector<ur_half4> texels(ur_dds_texel_count(&info));
    uint32_t reserved = 123;
    CHECK(ur_dds_decode_rgba16f(cube.data(), cube.size(), &info, texels.data(), &reserved) == UR_ASSET_OK && reserved == 0);
    // every truncation point of the header region, and a few inside the payload, must be rejected without touching memory
    for (size_t n = 0; n < 200; ++n) {
        std::vector<char> cut(cube.begin(), cube.begin() + (std::ptrdiff_t)n); // exact-size copy: reads past n are reports
        ur_dds_info i2{};
        CHECK(ur_dds_parse(cut.data(), cut.size(), &i2) != UR_ASSET_OK);
    }
    for (size_t n : {size_t(200), size_t(4000), cube.size() / 2, cube.size() - 1}) {
        std::vector<char> cut(cube.begin(), cube.begin() + (std::ptrdiff_t)n);
        ur_dds_info i2{};
        const int rc = ur_dds_parse(cut.data(), cut.size(), &i2);
        if (rc == UR_ASSET_OK) CHECK(ur_dds_decode_rgba16f(cut.data(), cut.size(), &i2, texels.data(), nullptr) != UR_ASSET_OK);
    }
    CHECK(ur_dds_parse(lut.data(), lut.size(), &info) == UR_ASSET_OK && info.width == 128 && info.height == 32);
    std::vector<uint16_t> rg(ur_dds_texel_count(&info) * 2);
    CHECK(ur_dds_copy_rg16(lut.data(), lut.size(), &info, rg.data()) == UR_ASSET_OK);
    CHECK(ur_dds_decode_rgba16f(lut.data(), lut.size(), &info, texels.data(), nullptr) != UR_ASSET_OK); // wrong format for this entry point
    // random blocks, both signednesses: every mode, reserved modes included (shift/overflow UB would be reported)
    std::mt19937 rng(7);
    for (int k = 0; k < 200000; ++k) {
        uint8_t b[16];
        for (auto& x : b) x = (uint8_t)rng();
        ur_half4 out[16];
        int32_t ep[12];
        ur_bc6h_decode_block(b, k & 1, out);
        const int mode = ur_bc6h_block_endpoints(b, k & 1, ep);
        CHECK(mode >= 0 && mode <= 14);
    }
}

static void test_scene(const std::string& assets)
{
    const char* scenes[3][2] = {{"Duck.json", "Duck/Duck.gltf"}, {"sponza.json", "sponza/untitled.gltf"}, {"pica_pica.json", "pica_pica/scene.gltf"}};
    const uint32_t expect[3] = {1, 25, 170};
    for (int s = 0; s < 3; ++s) {
        const std::string sj = slurp(assets + "/Scenes/" + scenes[s][0]), gj = slurp(assets + "/" + scenes[s][1]);
        if (sj.empty() || gj.empty()) { std::printf("note: %s not among the fixtures, skipped\n", scenes[s][0]); continue; }
        const char* g[1] = {gj.c_str()};
        ur_scene_summary sum{};
        CHECK(ur_scene_extract(sj.c_str(), g, 1, nullptr, 0, &sum) == UR_OK && sum.model_count == expect[s]);
        std::vector<ur_scene_model> models(sum.model_count); // exact size
        CHECK(ur_scene_extract(sj.c_str(), g, 1, models.data(), (uint32_t)models.size(), &sum) == UR_OK);
        if (!models.empty()) CHECK(ur_scene_extract(sj.c_str(), g, 1, models.data(), (uint32_t)models.size() - 1, &sum) == UR_SCENE_ECAPACITY);
        // hostile: every prefix of the glTF text at a coarse stride, and the scene JSON cut short
        for (size_t n = 0; n < gj.size(); n += gj.size() / 97 + 1) {
            const std::string cut = gj.substr(0, n);
            const char* gc[1] = {cut.c_str()};
            (void)ur_scene_extract(sj.c_str(), gc, 1, models.data(), (uint32_t)models.size(), &sum);
        }
        for (size_t n = 0; n < sj.size(); n += 7) {
            const std::string cut = sj.substr(0, n);
            (void)ur_scene_model_count(cut.c_str());
            (void)ur_scene_extract(cut.c_str(), g, 1, models.data(), (uint32_t)models.size(), &sum);
        }
        char buf[8];
        (void)ur_scene_model_path(sj.c_str(), 0, buf, sizeof buf); // too small a buffer must not be overrun
    }
}

static void fill_half4(std::vector<ur_half4>& v, std::mt19937& rng, float lo, float hi)
{
    std::uniform_real_distribution<float> d(lo, hi);
    for (auto& t : v) { t.x = uro_f2h(d(rng)); t.y = uro_f2h(d(rng)); t.z = uro_f2h(d(rng)); t.w = uro_f2h(d(rng) * 10.0f + 0.2f); }
}

static void test_oracle_and_host_math()
{
    std::mt19937 rng(11);
    uro_set_threads(3);
    const float eye[3] = {14.3f, 0.76f, 0.57f}, at[3] = {13.3f, 0.6f, 0.4f}, up[3] = {0, 1, 0};
    float view[16], proj[16], vp[16], lvp[16], planes[24];
    ur_host_look_at_lh(eye, at, up, view);
    ur_host_reverse_z_projection(1.0472f, 16.0f / 9.0f, 0.1f, proj);
    ur_host_mat_mul(view, proj, vp);
    ur_host_frustum_planes(vp, planes);
    const float center[3] = {0, 3, 0}, ldir[3] = {0.0f, 0.966f, 0.259f}, lcol[3] = {1, 1, 1};
    ur_host_light_view_projection(center, 30.0f, ldir, lvp);
    // sizes: 1x1, odd, ragged, and one whose mip chain takes three reference dispatches
    const uint32_t sizes[][2] = {{1, 1}, {2, 2}, {3, 5}, {17, 9}, {130, 66}, {257, 131}};
    for (auto& wh : sizes) {
        const uint32_t w = wh[0], h = wh[1];
        ur_mip_desc mips[UR_MAX_HZB_MIPS];
        uint32_t count = 0;
        const uint32_t total = uro_hzb_layout(w, h, mips, &count);
        CHECK(total > 0 && count >= 1);
        std::vector<float> depth((size_t)w * h), hzb(total); // exact sizes
        std::uniform_real_distribution<float> d01(0.0f, 1.0f);
        for (auto& x : depth) x = d01(rng) < 0.2f ? 0.0f : d01(rng) * 0.2f;
        uro_build_hzb(depth.data(), w, h, hzb.data(), mips, count);
        const uint32_t n = 257;
        std::vector<ur_float4> bounds(2 * n);
        std::uniform_real_distribution<float> pos(-40.0f, 40.0f), ext(0.05f, 5.0f);
        for (uint32_t i = 0; i < n; ++i) {
            const float c[3] = {eye[0] + pos(rng), eye[1] + pos(rng), eye[2] + pos(rng)}, e[3] = {ext(rng), ext(rng), ext(rng)};
            bounds[2 * i] = {c[0] - e[0], c[1] - e[1], c[2] - e[2], 0};
            bounds[2 * i + 1] = {c[0] + e[0], c[1] + e[1], c[2] + e[2], 0};
        }
        uint32_t consts[UR_CULL_CONSTANT_DWORDS];
        ur_host_pack_culling_constants(view, proj, n, 1, count, mips[0].width, mips[0].height, 1, consts);
        std::vector<uint32_t> args((size_t)n * 16, 1u), vis(n);
        uint32_t stats[2] = {0, 0}, nvis = 0;
        uro_cull_indirect_args(consts, bounds.data(), hzb.data(), mips, args.data(), stats, vis.data(), &nvis, 5);
        CHECK(nvis <= n && stats[0] + stats[1] + nvis == n);
        // lighting + sky + tonemap + TAA on the same frame size (band = the lower part of the frame: row0 > 0)
        ur_scene_constants S{};
        ur_sky_constants K{};
        ur_host_fill_scene_constants(view, proj, eye, 1.0f, ldir, lcol, lvp, 1.0f, 0.0f, 16.0f, 16.0f, 4.0f, &S);
        ur_host_fill_sky_constants(view, proj, eye, 5000.0f, ldir, lcol, &K);
        const uint32_t row0 = h / 3, rows = h - row0;
        std::vector<ur_half4> A((size_t)w * rows), B((size_t)w * rows), hdr((size_t)w * rows), hist((size_t)w * rows), taa((size_t)w * rows), full((size_t)w * h);
        std::vector<uint32_t> Cc((size_t)w * rows), ldr((size_t)w * rows);
        fill_half4(A, rng, -1.0f, 1.0f);
        fill_half4(B, rng, 0.0f, 1.0f);
        fill_half4(hdr, rng, 0.0f, 2.0f);
        fill_half4(hist, rng, 0.0f, 2.0f);
        fill_half4(full, rng, 0.0f, 2.0f);
        for (auto& x : Cc) x = (uint32_t)rng();
        std::vector<float> shadow(16 * 16);
        for (auto& x : shadow) x = d01(rng);
        const uint32_t base = 8, nm = 4;
        size_t src_texels = 0;
        for (uint32_t m = 0; m < nm; ++m) src_texels += 6u * (base >> m) * (base >> m);
        std::vector<ur_half4> cube(src_texels), staged(uro_env_cube_texels(base, nm));
        fill_half4(cube, rng, 0.0f, 3.0f);
        uro_stage_env_cube(cube.data(), base, nm, staged.data());
        std::vector<uint16_t> lut(16 * 8 * 2);
        for (auto& x : lut) x = (uint16_t)rng();
        std::vector<uint8_t> fragile((size_t)w * rows);
        uro_deferred_lighting(&S, A.data(), B.data(), Cc.data(), shadow.data(), cube.data(), base, nm, lut.data(), 16, 8, hdr.data(), w, h, row0, rows,
                              fragile.data());
        uro_sky_atmosphere(&K, depth.data() + (size_t)row0 * w, hdr.data(), w, h, row0, rows);
        ur_tonemap_constants TK{1, 0, 0.9f, 2.2f};
        uro_tonemap(&TK, hdr.data(), nullptr, ldr.data(), (uint32_t)hdr.size());
        uro_temporal_aa(full.data(), hist.data(), taa.data(), 0.9f, 1, w, h, row0, rows);
    }
}

int main(int argc, char** argv)
{
    const std::string assets = argc > 1 ? argv[1] : "tests/golden/assets";
    test_dds(assets);
    test_scene(assets);
    test_oracle_and_host_math();
    if (g_fail) { std::printf("%d check(s) failed\n", g_fail); return 1; }
    std::printf("OK sanitized host + oracle run clean\n");
    return 0;
}
