"""Per-kernel register / scratch figures of the shipped HIP sources, read from hipcc's own remarks.

``python tools/resource_usage.py [file.hip ...]`` compiles each translation unit for gfx950 with the
product's flags plus ``-Rpass-analysis=kernel-resource-usage`` (device pass only, no GPU needed) and
prints one line per kernel.  ``usage(src)`` returns the same as a list of dicts; the CPU suite
(tests/test_kernel_resources.py) fails on any hot kernel with scratch > 0 that is not allow-listed.
"""
from __future__ import annotations

import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from veritasfi_amd import build as vf_build  # noqa: E402

_FIELDS = {
    "Function Name": "name", "VGPRs": "vgprs", "AGPRs": "agprs", "ScratchSize [bytes/lane]": "scratch",
    "VGPRs Spill": "vgpr_spill", "SGPRs Spill": "sgpr_spill", "SGPRs": "sgprs", "Occupancy [waves/SIMD]": "occupancy",
    "LDS Size [bytes/block]": "lds", "Dynamic Stack": "dynamic_stack",
}
_LINE = re.compile(r"remark: (?:[^:]+:\d+:\d+: )?\s*([A-Za-z][A-Za-z /\[\]]*?):\s*(\S+)\s*\[-Rpass-analysis")


def demangle(names):
    """'_ZN3vft11k_gemm8p_tnILi0EEEv...' -> 'k_gemm8p_tn<0>' (the image's c++filt predates _Float16 manglings): the kernel's
    own identifier and its integral / bool template arguments are all a reader needs."""
    out = []
    for n in names:
        m, pos, ident = re.match(r"_ZN?", n), 0, None
        if not m:
            out.append(n)
            continue
        pos = m.end()
        while pos < len(n) and n[pos].isdigit():   # <len><identifier> components; the last one is the kernel
            ln = re.match(r"\d+", n[pos:])
            pos += ln.end()
            ident = n[pos:pos + int(ln.group())]
            pos += int(ln.group())
        targs = []
        if n[pos:pos + 1] == "I":
            for kind, neg, val in re.findall(r"L([a-z])(n?)(\d+)E", n[pos:n.index("EE", pos) + 1] if "EE" in n[pos:] else ""):
                targs.append(("true" if val == "1" else "false") if kind == "b" else ("-" if neg else "") + val)
        out.append((ident or n) + (f"<{','.join(targs)}>" if targs else ""))
    return out


def usage(src: str, extra=()):
    """[{name, pretty, vgprs, agprs, scratch, vgpr_spill, sgpr_spill, sgprs, occupancy, lds}] for every kernel of ``src``."""
    cmd = [vf_build._hipcc()] + vf_build.FLAGS + vf_build.EXTRA + list(extra) + [
        "--cuda-device-only", "-Rpass-analysis=kernel-resource-usage", "-c", src, "-o", os.devnull]
    proc = subprocess.run(cmd, capture_output=True, text=True, cwd=os.path.dirname(src))
    if proc.returncode != 0:
        raise RuntimeError(proc.stderr[-4000:])
    kernels, cur = [], None
    for line in proc.stderr.splitlines():
        m = _LINE.search(line)
        if not m:
            continue
        key = _FIELDS.get(m.group(1).strip())
        if key is None:
            continue
        if key == "name":
            cur = {"name": m.group(2)}
            kernels.append(cur)
        elif cur is not None:
            try:
                cur[key] = int(m.group(2))
            except ValueError:
                cur[key] = m.group(2)
    for k, p in zip(kernels, demangle([k["name"] for k in kernels])):
        k["pretty"] = p
    return kernels


def main(argv):
    srcs = argv or [os.path.join(vf_build.CSRC, s) for s in vf_build.SOURCES]
    for src in srcs:
        print(f"# {os.path.relpath(src, ROOT)}")
        print(f"{'kernel':60s} {'vgpr':>4s} {'agpr':>4s} {'scratch':>7s} {'vspill':>6s} {'sspill':>6s} {'occ':>3s}")
        for k in usage(os.path.abspath(src)):
            print(f"{k['pretty'][:60]:60s} {k.get('vgprs', 0):4d} {k.get('agprs', 0):4d} {k.get('scratch', 0):7d} "
                  f"{k.get('vgpr_spill', 0):6d} {k.get('sgpr_spill', 0):6d} {k.get('occupancy', 0):3d}")


if __name__ == "__main__":
    main(sys.argv[1:])
