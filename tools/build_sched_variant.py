#!/usr/bin/env python3
"""libur_<name>.so whose lighting kernels went through tools/valu_sched.py (device assembly -> post-pass -> assemble -> bundle -> host
compile with the bundle embedded). Usage: python tools/build_sched_variant.py name [valu_sched.py options ...]; name 'identity' with
option --none skips the post-pass (the round trip alone)."""
import subprocess
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
from unclerenderer_amd import build as b  # noqa: E402

LLVM = Path("/opt/rocm/lib/llvm/bin")


def sh(cmd):
    r = subprocess.run([str(c) for c in cmd], capture_output=True, text=True)
    if r.returncode != 0:
        raise SystemExit(f"failed: {' '.join(str(c) for c in cmd)}\n{r.stdout}\n{r.stderr}")
    return r


def build_lighting_object(out_dir: Path, name: str, sched_args: list[str]) -> Path:
    flags = dict(b.SOURCES)["lighting.hip"]
    src = b.CSRC / "lighting.hip"
    s0, s1 = out_dir / f"lighting_{name}.s", out_dir / f"lighting_{name}_sched.s"
    sh([b.hipcc()] + b.COMMON + flags + ["--cuda-device-only", "-S", src, "-o", s0])
    if "--none" in sched_args:
        s1.write_text(s0.read_text())
    else:
        r = sh([sys.executable, ROOT / "tools" / "valu_sched.py", s0, s1, "--report"] + sched_args)
        print(r.stderr.strip().splitlines()[-1])
    dev_o, hsaco, fb, obj = out_dir / f"lighting_{name}_dev.o", out_dir / f"lighting_{name}.hsaco", out_dir / f"lighting_{name}.hipfb", out_dir / f"lighting_{name}.o"
    sh([LLVM / "clang", "-x", "assembler", "-target", "amdgcn-amd-amdhsa", f"-mcpu={b.ARCH}", "-c", s1, "-o", dev_o])
    sh([LLVM / "lld", "-flavor", "gnu", "-m", "elf64_amdgpu", "--no-undefined", "-shared", "-o", hsaco, dev_o])
    sh([LLVM / "clang-offload-bundler", "-type=o", "-bundle-align=4096", f"-targets=host-x86_64-unknown-linux-gnu,hipv4-amdgcn-amd-amdhsa--{b.ARCH}",
        "-input=/dev/null", f"-input={hsaco}", f"-output={fb}"])
    sh([b.hipcc()] + b.COMMON + flags + ["--cuda-host-only", "-Xclang", "-fcuda-include-gpubinary", "-Xclang", fb, "-c", src, "-o", obj])
    for tmp in (s0, dev_o, hsaco, fb):
        tmp.unlink()
    return obj


def main():
    name, args = sys.argv[1], sys.argv[2:]
    b.build()
    out = b.OUT / "variants"
    out.mkdir(parents=True, exist_ok=True)
    obj = build_lighting_object(out, name, args)
    objs = [str(b.OUT / (s.replace("/", "_") + ".o")) for s, _ in b.SOURCES if s != "lighting.hip"] + [str(obj)]
    lib = out / f"libur_{name}.so"
    sh([b.hipcc(), f"--offload-arch={b.ARCH}", "-shared", "-fPIC", "-o", lib] + objs + ["-ldl", "-lpthread"])
    obj.unlink()
    print(lib)


if __name__ == "__main__":
    main()
