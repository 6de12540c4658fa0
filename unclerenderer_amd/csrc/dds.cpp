// DDS container parse + BC6H (UF16/SF16) block decode to RGBA16F, and RG16 pass-through — the on-disk formats of the
// lighting pass's IBL inputs: Assets/Textures/output_pmrem.dds (DX10 header, DXGI_FORMAT_BC6H_SF16,
// cube, 256^2, 9 mips) and Assets/Textures/PreintegratedGF.dds (legacy header, 32 bpp masks 0xffff/0xffff0000 =
// R16G16_UNORM, 128x32). The reference hands the compressed blocks to D3D12 and the texture unit decodes them
// (Source/Render/TextureLoader.cpp:178-315 walks the file slice-major, mips inner; DeferredRenderer.cpp:306-330 loads
// the two files); CDNA has no block-compression hardware, so the chain is decoded ONCE at setup on the host and staged
// with ur_stage_env_cube(). Written from the published BC6H format description (Microsoft "BC6H Format", Khronos Data
// Format Specification §BPTC); PARITY UNPINNED: no decoder or decoded image ships with the reference.

#include <cmath>
#include <cstdint>
#include <cstring>

#include "../../include/ur_assets.h"

namespace {

constexpr uint32_t kMagic = 0x20534444u; // "DDS "
constexpr uint32_t kFourCC_DX10 = 0x30315844u;
constexpr uint32_t DDPF_FOURCC = 0x4u, DDPF_RGB = 0x40u;
constexpr uint32_t DDSCAPS2_CUBEMAP = 0x200u;
constexpr uint32_t DXGI_R16G16B16A16_FLOAT = 10, DXGI_R16G16_UNORM = 35, DXGI_BC6H_UF16 = 95, DXGI_BC6H_SF16 = 96;

uint32_t rd32(const uint8_t* p) { return (uint32_t)p[0] | ((uint32_t)p[1] << 8) | ((uint32_t)p[2] << 16) | ((uint32_t)p[3] << 24); }

// ---- BC6H -----------------------------------------------------------------------------------------------------------------
struct BitReader {
    uint64_t lo, hi;
    uint32_t pos = 0;
    uint32_t bit()
    {
        const uint32_t b = pos < 64 ? (uint32_t)((lo >> pos) & 1u) : (uint32_t)((hi >> (pos - 64)) & 1u);
        ++pos;
        return b;
    }
    uint32_t bits(int n) // LSB first
    {
        uint32_t v = 0;
        for (int i = 0; i < n; ++i) v |= bit() << i;
        return v;
    }
    uint32_t bits_rev(int n) // first bit read is the MOST significant (the spec's "[10:15]" fields)
    {
        uint32_t v = 0;
        for (int i = 0; i < n; ++i) v = (v << 1) | bit();
        return v;
    }
};

// Two-subset partition table (first 32 shapes of the BPTC set) and the anchor (fix-up) index of subset 1.
const uint8_t kPartition2[32][16] = {
    {0, 0, 1, 1, 0, 0, 1, 1, 0, 0, 1, 1, 0, 0, 1, 1}, {0, 0, 0, 1, 0, 0, 0, 1, 0, 0, 0, 1, 0, 0, 0, 1}, {0, 1, 1, 1, 0, 1, 1, 1, 0, 1, 1, 1, 0, 1, 1, 1},
    {0, 0, 0, 1, 0, 0, 1, 1, 0, 0, 1, 1, 0, 1, 1, 1}, {0, 0, 0, 0, 0, 0, 0, 1, 0, 0, 0, 1, 0, 0, 1, 1}, {0, 0, 1, 1, 0, 1, 1, 1, 0, 1, 1, 1, 1, 1, 1, 1},
    {0, 0, 0, 1, 0, 0, 1, 1, 0, 1, 1, 1, 1, 1, 1, 1}, {0, 0, 0, 0, 0, 0, 0, 1, 0, 0, 1, 1, 0, 1, 1, 1}, {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 1, 0, 0, 1, 1},
    {0, 0, 1, 1, 0, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1}, {0, 0, 0, 0, 0, 0, 0, 1, 0, 1, 1, 1, 1, 1, 1, 1}, {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 1, 0, 1, 1, 1},
    {0, 0, 0, 1, 0, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1}, {0, 0, 0, 0, 0, 0, 0, 0, 1, 1, 1, 1, 1, 1, 1, 1}, {0, 0, 0, 0, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1},
    {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 1, 1, 1, 1}, {0, 0, 0, 0, 1, 0, 0, 0, 1, 1, 1, 0, 1, 1, 1, 1}, {0, 1, 1, 1, 0, 0, 0, 1, 0, 0, 0, 0, 0, 0, 0, 0},
    {0, 0, 0, 0, 0, 0, 0, 0, 1, 0, 0, 0, 1, 1, 1, 0}, {0, 1, 1, 1, 0, 0, 1, 1, 0, 0, 0, 1, 0, 0, 0, 0}, {0, 0, 1, 1, 0, 0, 0, 1, 0, 0, 0, 0, 0, 0, 0, 0},
    {0, 0, 0, 0, 1, 0, 0, 0, 1, 1, 0, 0, 1, 1, 1, 0}, {0, 0, 0, 0, 0, 0, 0, 0, 1, 0, 0, 0, 1, 1, 0, 0}, {0, 1, 1, 1, 0, 0, 1, 1, 0, 0, 1, 1, 0, 0, 0, 1},
    {0, 0, 1, 1, 0, 0, 0, 1, 0, 0, 0, 1, 0, 0, 0, 0}, {0, 0, 0, 0, 1, 0, 0, 0, 1, 0, 0, 0, 1, 1, 0, 0}, {0, 1, 1, 0, 0, 1, 1, 0, 0, 1, 1, 0, 0, 1, 1, 0},
    {0, 0, 1, 1, 0, 1, 1, 0, 0, 1, 1, 0, 1, 1, 0, 0}, {0, 0, 0, 1, 0, 1, 1, 1, 1, 1, 1, 0, 1, 0, 0, 0}, {0, 0, 0, 0, 1, 1, 1, 1, 1, 1, 1, 1, 0, 0, 0, 0},
    {0, 1, 1, 1, 0, 0, 0, 1, 1, 0, 0, 0, 1, 1, 1, 0}, {0, 0, 1, 1, 1, 0, 0, 1, 1, 0, 0, 1, 1, 1, 0, 0}};
const uint8_t kAnchor2[32] = {15, 15, 15, 15, 15, 15, 15, 15, 15, 15, 15, 15, 15, 15, 15, 15, 15, 2, 8, 2, 2, 8, 8, 15, 2, 8, 2, 2, 8, 8, 2, 2};
const int kWeights3[8] = {0, 9, 18, 27, 37, 46, 55, 64};
const int kWeights4[16] = {0, 4, 9, 13, 17, 21, 26, 30, 34, 38, 43, 47, 51, 55, 60, 64};

struct ModeInfo { int regions, transformed, wbits, dr, dg, db; };

int sign_extend(int v, int bits)
{
    const int m = 1 << (bits - 1);
    v &= (1 << bits) - 1;
    return (v ^ m) - m;
}

int unquantize(int x, int bits, bool is_signed)
{
    if (is_signed) {
        if (bits >= 16) return x;
        const bool neg = x < 0;
        if (neg) x = -x;
        int unq;
        if (x == 0) unq = 0;
        else if (x >= ((1 << (bits - 1)) - 1)) unq = 0x7FFF;
        else unq = ((x << 15) + 0x4000) >> (bits - 1);
        return neg ? -unq : unq;
    }
    if (bits >= 15) return x;
    if (x == 0) return 0;
    if (x == ((1 << bits) - 1)) return 0xFFFF;
    return ((x << 15) + 0x4000) >> (bits - 1);
}

uint16_t finish_unquantize(int v, bool is_signed)
{
    if (is_signed) {
        const bool neg = v < 0;
        if (neg) v = -v;
        v = (v * 31) >> 5;
        return (uint16_t)(neg ? (0x8000 | v) : v);
    }
    return (uint16_t)((v * 31) >> 6);
}

// Decode one 16-byte block into 16 texels (row-major 4x4), alpha = 1.0. Returns false on a reserved mode.
// endpoints_out (nullable): the unquantized endpoints r[4], g[4], b[4] (subset 0: [0],[1]; subset 1: [2],[3]); mode_out
// (nullable): the mode number 1..14 (0 = reserved).
bool decode_bc6h_block(const uint8_t* block, bool is_signed, ur_half4 out[16], int32_t* endpoints_out = nullptr, int* mode_out = nullptr)
{
    BitReader br;
    std::memcpy(&br.lo, block, 8);
    std::memcpy(&br.hi, block + 8, 8);
    int r[4] = {0, 0, 0, 0}, g[4] = {0, 0, 0, 0}, b[4] = {0, 0, 0, 0};
    uint32_t partition = 0;
    uint32_t mode = br.bits(2);
    if (mode > 1) mode |= br.bits(3) << 2;
    ModeInfo mi;
#define RW(n) r[0] |= br.bits(n)
#define GW(n) g[0] |= br.bits(n)
#define BW(n) b[0] |= br.bits(n)
#define RX(n) r[1] |= br.bits(n)
#define GX(n) g[1] |= br.bits(n)
#define BX(n) b[1] |= br.bits(n)
#define RY(n) r[2] |= br.bits(n)
#define GY(n) g[2] |= br.bits(n)
#define BY(n) b[2] |= br.bits(n)
#define RZ(n) r[3] |= br.bits(n)
#define GZ(n) g[3] |= br.bits(n)
#define BZ(n) b[3] |= br.bits(n)
#define GYb(k) g[2] |= br.bit() << (k)
#define BYb(k) b[2] |= br.bit() << (k)
#define GZb(k) g[3] |= br.bit() << (k)
#define BZb(k) b[3] |= br.bit() << (k)
#define RWb(k) r[0] |= br.bit() << (k)
#define GWb(k) g[0] |= br.bit() << (k)
#define BWb(k) b[0] |= br.bit() << (k)
    switch (mode) {
    case 0x00: // mode 1: 10.555
        GYb(4); BYb(4); BZb(4); RW(10); GW(10); BW(10); RX(5); GZb(4); GY(4); GX(5); BZb(0); GZ(4); BX(5); BZb(1); BY(4); RY(5); BZb(2); RZ(5); BZb(3);
        partition = br.bits(5);
        mi = {2, 1, 10, 5, 5, 5};
        break;
    case 0x01: // mode 2: 7.666
        GYb(5); GZb(4); GZb(5); RW(7); BZb(0); BZb(1); BYb(4); GW(7); BYb(5); BZb(2); GYb(4); BW(7); BZb(3); BZb(5); BZb(4); RX(6); GY(4); GX(6); GZ(4); BX(6);
        BY(4); RY(6); RZ(6);
        partition = br.bits(5);
        mi = {2, 1, 7, 6, 6, 6};
        break;
    case 0x02: // mode 3: 11.544
        RW(10); GW(10); BW(10); RX(5); RWb(10); GY(4); GX(4); GWb(10); BZb(0); GZ(4); BX(4); BWb(10); BZb(1); BY(4); RY(5); BZb(2); RZ(5); BZb(3);
        partition = br.bits(5);
        mi = {2, 1, 11, 5, 4, 4};
        break;
    case 0x06: // mode 4: 11.454
        RW(10); GW(10); BW(10); RX(4); RWb(10); GZb(4); GY(4); GX(5); GWb(10); GZ(4); BX(4); BWb(10); BZb(1); BY(4); RY(4); BZb(0); BZb(2); RZ(4); GYb(4); BZb(3);
        partition = br.bits(5);
        mi = {2, 1, 11, 4, 5, 4};
        break;
    case 0x0A: // mode 5: 11.445
        RW(10); GW(10); BW(10); RX(4); RWb(10); BYb(4); GY(4); GX(4); GWb(10); BZb(0); GZ(4); BX(5); BWb(10); BY(4); RY(4); BZb(1); BZb(2); RZ(4); BZb(4); BZb(3);
        partition = br.bits(5);
        mi = {2, 1, 11, 4, 4, 5};
        break;
    case 0x0E: // mode 6: 9.555
        RW(9); BYb(4); GW(9); GYb(4); BW(9); BZb(4); RX(5); GZb(4); GY(4); GX(5); BZb(0); GZ(4); BX(5); BZb(1); BY(4); RY(5); BZb(2); RZ(5); BZb(3);
        partition = br.bits(5);
        mi = {2, 1, 9, 5, 5, 5};
        break;
    case 0x12: // mode 7: 8.655
        RW(8); GZb(4); BYb(4); GW(8); BZb(2); GYb(4); BW(8); BZb(3); BZb(4); RX(6); GY(4); GX(5); BZb(0); GZ(4); BX(5); BZb(1); BY(4); RY(6); RZ(6);
        partition = br.bits(5);
        mi = {2, 1, 8, 6, 5, 5};
        break;
    case 0x16: // mode 8: 8.565
        RW(8); BZb(0); BYb(4); GW(8); GYb(5); GYb(4); BW(8); GZb(5); BZb(4); RX(5); GZb(4); GY(4); GX(6); GZ(4); BX(5); BZb(1); BY(4); RY(5); BZb(2); RZ(5); BZb(3);
        partition = br.bits(5);
        mi = {2, 1, 8, 5, 6, 5};
        break;
    case 0x1A: // mode 9: 8.556
        RW(8); BZb(1); BYb(4); GW(8); BYb(5); GYb(4); BW(8); BZb(5); BZb(4); RX(5); GZb(4); GY(4); GX(5); BZb(0); GZ(4); BX(6); BY(4); RY(5); BZb(2); RZ(5); BZb(3);
        partition = br.bits(5);
        mi = {2, 1, 8, 5, 5, 6};
        break;
    case 0x1E: // mode 10: 6.666, endpoints stored directly
        RW(6); GZb(4); BZb(0); BZb(1); BYb(4); GW(6); GYb(5); BYb(5); BZb(2); GYb(4); BW(6); GZb(5); BZb(3); BZb(5); BZb(4); RX(6); GY(4); GX(6); GZ(4); BX(6);
        BY(4); RY(6); RZ(6);
        partition = br.bits(5);
        mi = {2, 0, 6, 6, 6, 6};
        break;
    case 0x03: // mode 11: 10.10, one region, direct
        RW(10); GW(10); BW(10); RX(10); GX(10); BX(10);
        mi = {1, 0, 10, 10, 10, 10};
        break;
    case 0x07: // mode 12: 11.9
        RW(10); GW(10); BW(10); RX(9); RWb(10); GX(9); GWb(10); BX(9); BWb(10);
        mi = {1, 1, 11, 9, 9, 9};
        break;
    case 0x0B: // mode 13: 12.8
        RW(10); GW(10); BW(10); RX(8); r[0] |= br.bits_rev(2) << 10; GX(8); g[0] |= br.bits_rev(2) << 10; BX(8); b[0] |= br.bits_rev(2) << 10;
        mi = {1, 1, 12, 8, 8, 8};
        break;
    case 0x0F: // mode 14: 16.4
        RW(10); GW(10); BW(10); RX(4); r[0] |= br.bits_rev(6) << 10; GX(4); g[0] |= br.bits_rev(6) << 10; BX(4); b[0] |= br.bits_rev(6) << 10;
        mi = {1, 1, 16, 4, 4, 4};
        break;
    default: // reserved modes decode to zero (D3D behaviour)
        for (int i = 0; i < 16; ++i) out[i] = {0, 0, 0, 0x3C00};
        if (mode_out) *mode_out = 0;
        if (endpoints_out) std::memset(endpoints_out, 0, 12 * sizeof(int32_t));
        return false;
    }
    if (mode_out) {
        static const uint8_t kModeNumber[32] = {1, 2, 3, 11, 0, 0, 4, 12, 0, 0, 5, 13, 0, 0, 6, 14, 0, 0, 7, 0, 0, 0, 8, 0, 0, 0, 9, 0, 0, 0, 10, 0};
        *mode_out = kModeNumber[mode];
    }
#undef RW
#undef GW
#undef BW
#undef RX
#undef GX
#undef BX
#undef RY
#undef GY
#undef BY
#undef RZ
#undef GZ
#undef BZ
#undef GYb
#undef BYb
#undef GZb
#undef BZb
#undef RWb
#undef GWb
#undef BWb
    const int num_ep = mi.regions * 2;
    int* ch[3] = {r, g, b};
    const int dbits[3] = {mi.dr, mi.dg, mi.db};
    for (int c = 0; c < 3; ++c) {
        int* e = ch[c];
        if (is_signed) e[0] = sign_extend(e[0], mi.wbits);
        if (mi.transformed) {
            for (int i = 1; i < num_ep; ++i) {
                e[i] = (e[0] + sign_extend(e[i], dbits[c])) & ((1 << mi.wbits) - 1);
                if (is_signed) e[i] = sign_extend(e[i], mi.wbits);
            }
        } else if (is_signed) {
            for (int i = 1; i < num_ep; ++i) e[i] = sign_extend(e[i], mi.wbits); // direct modes: delta width == endpoint width
        }
        if (!is_signed) e[0] &= (1 << mi.wbits) - 1;
        for (int i = 0; i < num_ep; ++i) e[i] = unquantize(e[i], mi.wbits, is_signed);
    }
    if (endpoints_out)
        for (int i = 0; i < 4; ++i) { endpoints_out[i] = r[i]; endpoints_out[4 + i] = g[i]; endpoints_out[8 + i] = b[i]; }
    const int ibits = mi.regions == 2 ? 3 : 4;
    const int* weights = mi.regions == 2 ? kWeights3 : kWeights4;
    for (int i = 0; i < 16; ++i) {
        const int subset = mi.regions == 2 ? kPartition2[partition][i] : 0;
        const bool anchor = (i == 0) || (mi.regions == 2 && i == kAnchor2[partition]);
        const int idx = (int)br.bits(anchor ? ibits - 1 : ibits);
        const int w = weights[idx];
        const int e0 = subset * 2, e1 = e0 + 1;
        const uint16_t hr = finish_unquantize((r[e0] * (64 - w) + r[e1] * w + 32) >> 6, is_signed);
        const uint16_t hg = finish_unquantize((g[e0] * (64 - w) + g[e1] * w + 32) >> 6, is_signed);
        const uint16_t hb = finish_unquantize((b[e0] * (64 - w) + b[e1] * w + 32) >> 6, is_signed);
        out[i] = {hr, hg, hb, 0x3C00};
    }
    return true;
}

uint32_t mip_dim(uint32_t base, uint32_t m) { return (base >> m) > 1u ? (base >> m) : 1u; }

} // namespace

extern "C" {

int ur_dds_parse(const void* file, size_t size, ur_dds_info* out)
{
    if (!file || !out || size < 128) return UR_ASSET_EINVAL;
    const uint8_t* p = static_cast<const uint8_t*>(file);
    if (rd32(p) != kMagic || rd32(p + 4) != 124) return UR_ASSET_EFORMAT;
    ur_dds_info d{};
    d.height = rd32(p + 12);
    d.width = rd32(p + 16);
    d.mip_count = rd32(p + 28) ? rd32(p + 28) : 1u;
    const uint32_t pf_flags = rd32(p + 80), fourcc = rd32(p + 84), rgb_bits = rd32(p + 88);
    const uint32_t rmask = rd32(p + 92), gmask = rd32(p + 96), bmask = rd32(p + 100), amask = rd32(p + 104);
    const uint32_t caps2 = rd32(p + 112);
    d.header_size = 128;
    d.slices = 1;
    d.is_cube = (caps2 & DDSCAPS2_CUBEMAP) ? 1u : 0u;
    if ((pf_flags & DDPF_FOURCC) && fourcc == kFourCC_DX10) {
        if (size < 148) return UR_ASSET_EFORMAT;
        d.header_size = 148;
        d.dxgi_format = rd32(p + 128);
        const uint32_t misc = rd32(p + 136), array_size = rd32(p + 140);
        if (misc & 0x4u) d.is_cube = 1u; // D3D10_RESOURCE_MISC_TEXTURECUBE
        d.slices = array_size ? array_size : 1u;
    } else if ((pf_flags & DDPF_RGB) && rgb_bits == 32 && rmask == 0x0000FFFFu && gmask == 0xFFFF0000u && bmask == 0 && amask == 0) {
        d.dxgi_format = DXGI_R16G16_UNORM;
    } else if ((pf_flags & DDPF_FOURCC) && fourcc == 113u) {
        d.dxgi_format = DXGI_R16G16B16A16_FLOAT; // D3DFMT_A16B16G16R16F
    } else {
        return UR_ASSET_EUNSUPPORTED;
    }
    if (d.is_cube) d.slices *= 6u;
    switch (d.dxgi_format) {
    case DXGI_BC6H_UF16:
    case DXGI_BC6H_SF16: d.block_dim = 4; d.bytes_per_block = 16; break;
    case DXGI_R16G16B16A16_FLOAT: d.block_dim = 1; d.bytes_per_block = 8; break;
    case DXGI_R16G16_UNORM: d.block_dim = 1; d.bytes_per_block = 4; break;
    default: return UR_ASSET_EUNSUPPORTED;
    }
    if (d.width == 0 || d.height == 0 || d.mip_count > 16) return UR_ASSET_EFORMAT;
    // the file must hold every subresource: slices outer, mips inner (TextureLoader.cpp:276-315)
    size_t need = d.header_size;
    for (uint32_t m = 0; m < d.mip_count; ++m) {
        const size_t bw = (mip_dim(d.width, m) + d.block_dim - 1) / d.block_dim, bh = (mip_dim(d.height, m) + d.block_dim - 1) / d.block_dim;
        need += (size_t)d.slices * bw * bh * d.bytes_per_block;
    }
    if (need > size) return UR_ASSET_EFORMAT;
    *out = d;
    return UR_ASSET_OK;
}

size_t ur_dds_texel_count(const ur_dds_info* d)
{
    if (!d) return 0;
    size_t n = 0;
    for (uint32_t m = 0; m < d->mip_count; ++m) n += (size_t)mip_dim(d->width, m) * mip_dim(d->height, m);
    return n * d->slices;
}

int ur_dds_decode_rgba16f(const void* file, size_t size, const ur_dds_info* d, ur_half4* out, uint32_t* reserved_blocks)
{
    if (!file || !d || !out || size < d->header_size) return UR_ASSET_EINVAL;
    const bool bc6 = d->dxgi_format == DXGI_BC6H_UF16 || d->dxgi_format == DXGI_BC6H_SF16;
    if (!bc6 && d->dxgi_format != DXGI_R16G16B16A16_FLOAT) return UR_ASSET_EUNSUPPORTED;
    const bool is_signed = d->dxgi_format == DXGI_BC6H_SF16;
    const uint8_t* src = static_cast<const uint8_t*>(file) + d->header_size;
    uint32_t bad = 0;
    for (uint32_t s = 0; s < d->slices; ++s)
        for (uint32_t m = 0; m < d->mip_count; ++m) {
            const uint32_t w = mip_dim(d->width, m), h = mip_dim(d->height, m);
            if (!bc6) {
                std::memcpy(out, src, (size_t)w * h * 8);
                src += (size_t)w * h * 8;
            } else {
                const uint32_t bw = (w + 3) / 4, bh = (h + 3) / 4;
                for (uint32_t by = 0; by < bh; ++by)
                    for (uint32_t bx = 0; bx < bw; ++bx) {
                        ur_half4 texels[16];
                        if (!decode_bc6h_block(src, is_signed, texels)) ++bad;
                        src += 16;
                        for (uint32_t ty = 0; ty < 4; ++ty)
                            for (uint32_t tx = 0; tx < 4; ++tx) {
                                const uint32_t x = bx * 4 + tx, y = by * 4 + ty;
                                if (x < w && y < h) out[(size_t)y * w + x] = texels[ty * 4 + tx];
                            }
                    }
            }
            out += (size_t)w * h;
        }
    if (reserved_blocks) *reserved_blocks = bad;
    return UR_ASSET_OK;
}

int ur_dds_copy_rg16(const void* file, size_t size, const ur_dds_info* d, uint16_t* out)
{
    if (!file || !d || !out) return UR_ASSET_EINVAL;
    if (d->dxgi_format != DXGI_R16G16_UNORM) return UR_ASSET_EUNSUPPORTED;
    const size_t bytes = ur_dds_texel_count(d) * 4;
    if (d->header_size + bytes > size) return UR_ASSET_EFORMAT;
    std::memcpy(out, static_cast<const uint8_t*>(file) + d->header_size, bytes);
    return UR_ASSET_OK;
}

int ur_bc6h_decode_block(const uint8_t block[16], int is_signed, ur_half4 out[16]) { return decode_bc6h_block(block, is_signed != 0, out) ? 1 : 0; }

int ur_bc6h_block_endpoints(const uint8_t block[16], int is_signed, int32_t endpoints[12])
{
    ur_half4 texels[16];
    int mode = 0;
    decode_bc6h_block(block, is_signed != 0, texels, endpoints, &mode);
    return mode;
}

} // extern "C"
