#!/usr/bin/env python3
"""Mint tests/golden/bc6h_pillow.npz: BC6H blocks of every one of the 14 modes, decoded by an INDEPENDENT third-party
implementation (Pillow's DDS plugin / BcnDecode.c). The reference decodes BC6H in D3D12 hardware (Source/Render/
TextureLoader.cpp:178-315) and ships no decoder or decoded image, so Pillow is the one external pin available offline.

Pillow returns 8-bit RGB: clamp(value, 0, 1) * 255 truncated. Blocks are drawn so that most texels fall inside (0, 1):
base endpoints in the lower half of their range, random deltas / partitions / indices; structured blocks (one field's one
bit set at a time) are added per mode so that every bit of every endpoint field is exercised in isolation.

    python tests/golden/make_bc6h_golden.py        # needs Pillow; writes the npz next to this script
Also writes the Pillow decode of the shipped cube's face +X, mip 0 (pmrem_face0_pillow_u8).
"""
import io
import struct
import sys
from pathlib import Path

import numpy as np

HERE = Path(__file__).resolve().parent
MODE_BITS = [(0x00, 2), (0x01, 2), (0x02, 5), (0x06, 5), (0x0A, 5), (0x0E, 5), (0x12, 5), (0x16, 5), (0x1A, 5), (0x1E, 5), (0x03, 5), (0x07, 5), (0x0B, 5), (0x0F, 5)]


def dds_bc6h(blocks: np.ndarray, bw: int, bh: int, signed: bool) -> bytes:
    """A DX10-header DDS of bw x bh blocks (row-major)."""
    assert blocks.shape == (bw * bh, 16)
    w, h = 4 * bw, 4 * bh
    hdr = struct.pack("<4sIIIIIII44xIIIIIIIIIIII4x", b"DDS ", 124, 0x1 | 0x2 | 0x4 | 0x1000 | 0x80000, h, w, bw * bh * 16, 0, 1,
                      32, 0x4, 0x30315844, 0, 0, 0, 0, 0, 0x1000, 0, 0, 0)
    assert len(hdr) == 128, len(hdr)
    dx10 = struct.pack("<IIIII", 96 if signed else 95, 3, 0, 1, 0)
    return hdr + dx10 + blocks.tobytes()


def pillow_decode(data: bytes) -> np.ndarray:
    from PIL import Image
    im = Image.open(io.BytesIO(data))
    im.load()
    return np.asarray(im.convert("RGB"))


def random_blocks(rng, per_mode: int) -> tuple[np.ndarray, np.ndarray]:
    out, modes = [], []
    for mi, (mode, nbits) in enumerate(MODE_BITS):
        for k in range(per_mode):
            v = int.from_bytes(rng.bytes(16), "little")
            if k % 4 != 3:
                # keep the base endpoint's top bits clear in most blocks so that texels land inside (0, 1): rw/gw/bw's ten
                # low bits sit at [5,35) in every mode but 2, 6, 7-9, 10 (narrower fields, same start or interleaved)
                for pos in (13, 14, 23, 24, 33, 34):
                    v &= ~(1 << pos)
            v = (v & ~((1 << nbits) - 1)) | mode
            out.append(v.to_bytes(16, "little"))
            modes.append(mi + 1)
        # structured: exactly one payload bit set (every bit position of the block in turn, indices included)
        for pos in range(nbits, 128):
            out.append((mode | (1 << pos)).to_bytes(16, "little"))
            modes.append(mi + 1)
        # and one payload bit set on top of mid-grey bases (deltas then move a visible value)
        for pos in range(nbits, 82):
            base = mode | (0x155 << 5) | (0x0AA << 15) | (0x133 << 25)
            out.append((base ^ (1 << pos)).to_bytes(16, "little"))
            modes.append(mi + 1)
    return np.frombuffer(b"".join(out), np.uint8).reshape(-1, 16).copy(), np.array(modes, np.uint8)


def untile(img: np.ndarray, bw: int, n: int) -> np.ndarray:
    """(4 bh, 4 bw, 3) image -> (n, 16, 3) per-block texels, row-major inside a block."""
    bh = img.shape[0] // 4
    t = img.reshape(bh, 4, bw, 4, 3).transpose(0, 2, 1, 3, 4).reshape(bh * bw, 16, 3)
    return t[:n]


def main():
    rng = np.random.default_rng(0xBC6)
    blocks, modes = random_blocks(rng, 192)
    n = blocks.shape[0]
    bw = 64
    bh = (n + bw - 1) // bw
    padded = np.zeros((bw * bh, 16), np.uint8)
    padded[:n] = blocks
    padded[n:, 0] = 3  # mode 11, all zero
    res = {}
    for signed in (False, True):
        img = pillow_decode(dds_bc6h(padded, bw, bh, signed))
        res["sf16" if signed else "uf16"] = untile(img, bw, n)
    face0 = pillow_decode((HERE / "assets" / "output_pmrem.dds").read_bytes())
    import PIL
    np.savez_compressed(HERE / "bc6h_pillow.npz", blocks=blocks, modes=modes, uf16=res["uf16"], sf16=res["sf16"], pmrem_face0_pillow_u8=face0,
                        pillow_version=np.array(PIL.__version__))
    print(f"{n} blocks, modes {np.bincount(modes)[1:]}, face0 {face0.shape}, Pillow {PIL.__version__}")


if __name__ == "__main__":
    sys.exit(main())
