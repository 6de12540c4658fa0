/*
 * ur_assets.h — setup-time staging of the lighting pass's IBL inputs from their on-disk format:
 * DDS container parse (DX10 and legacy headers), BC6H_UF16 / BC6H_SF16 block decode to RGBA16F, R16G16_UNORM copy.
 * Host-only, no GPU. Replaces what the reference gets from D3D12 + ddspp (Source/Render/TextureLoader.cpp:178-315;
 * loads at DeferredRenderer.cpp:306-330): the decoded chain is in the same order the loader walks the file
 * (slice-major, mips inner) — exactly what ur_stage_env_cube() takes.
 */
#ifndef UR_ASSETS_H
#define UR_ASSETS_H

#include <stddef.h>
#include <stdint.h>

#include "ur_hotpath.h"

#ifdef __cplusplus
extern "C" {
#endif

#define UR_ASSET_OK 0
#define UR_ASSET_EINVAL (-1)
#define UR_ASSET_EFORMAT (-2)      /* not a DDS file / truncated */
#define UR_ASSET_EUNSUPPORTED (-3) /* a pixel format other than BC6H, RGBA16F, RG16 */

typedef struct ur_dds_info {
    uint32_t width, height, mip_count;
    uint32_t slices;          /* array size x 6 for cubes */
    uint32_t is_cube;
    uint32_t dxgi_format;     /* 95 BC6H_UF16, 96 BC6H_SF16, 10 R16G16B16A16_FLOAT, 35 R16G16_UNORM */
    uint32_t header_size;     /* 128, or 148 with the DX10 extension */
    uint32_t block_dim;       /* 4 for BC6H, 1 otherwise */
    uint32_t bytes_per_block; /* 16 / 8 / 4 */
} ur_dds_info;

int ur_dds_parse(const void* file, size_t size, ur_dds_info* out);
/* texels of the whole chain (all slices, all mips) */
size_t ur_dds_texel_count(const ur_dds_info* info);
/* BC6H / RGBA16F file -> RGBA16F texels, slice-major, mips inner, rows top-down. reserved_blocks (nullable) counts BC6H
 * blocks with a reserved mode (decoded as zero, like D3D). */
int ur_dds_decode_rgba16f(const void* file, size_t size, const ur_dds_info* info, ur_half4* out, uint32_t* reserved_blocks);
/* R16G16_UNORM file -> 2 x uint16 per texel */
int ur_dds_copy_rg16(const void* file, size_t size, const ur_dds_info* info, uint16_t* out);
/* one BC6H block -> 16 texels (row-major 4x4); returns 0 for a reserved mode */
int ur_bc6h_decode_block(const uint8_t block[16], int is_signed, ur_half4 out[16]);
/* the block's mode number (1..14, 0 = reserved) and its unquantized endpoints r[4], g[4], b[4] (subset 0: [0],[1]; subset 1:
 * [2],[3]; one-region modes leave [2],[3] zero) — what the decoder interpolates between; for tests and asset diagnostics */
int ur_bc6h_block_endpoints(const uint8_t block[16], int is_signed, int32_t endpoints[12]);

#ifdef __cplusplus
}
#endif
#endif /* UR_ASSETS_H */
