#!/usr/bin/env python3
"""Build-time check of the gfx950 code objects in csrc/*.o: per kernel VGPRs / SGPRs / LDS / scratch from the code-object
notes (llvm-readelf --notes), and a non-zero exit when any kernel has a private (scratch) segment or spilled registers.

    python tools/check_codeobj.py            # table of offenders only; exit 1 if there are any
    python tools/check_codeobj.py --all      # every kernel
    python tools/check_codeobj.py --md       # markdown table (profiles/rNN_kernel_resources.md)

A kernel with scratch pays a few microseconds on each side of its dispatch on this stack -- for the 5-30 us
kernels of this library that is a double-digit percentage, so __graft_entry__.build() runs this check."""
import argparse
import re
import subprocess
import sys
import tempfile
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
CSRC = ROOT / "wakeword_trainer_home_amd" / "csrc"
LLVM = Path("/opt/rocm/lib/llvm/bin")
TARGET = "hipv4-amdgcn-amd-amdhsa--gfx950"
FIELDS = (".name", ".vgpr_count", ".agpr_count", ".sgpr_count", ".group_segment_fixed_size", ".private_segment_fixed_size",
          ".vgpr_spill_count", ".sgpr_spill_count", ".max_flat_workgroup_size")


def demangle(names):
    try:
        out = subprocess.run([str(LLVM / "llvm-cxxfilt")], input="\n".join(names), capture_output=True, text=True,
                             check=True).stdout.splitlines()
        return out if len(out) == len(names) else names
    except Exception:
        return names


def kernels_of(obj: Path, tmp: Path):
    co, fat = tmp / (obj.stem + ".co"), tmp / (obj.stem + ".fatbin")
    # the host object carries the device code as an offload bundle in its .hip_fatbin section
    subprocess.run([str(LLVM / "llvm-objcopy"), "--dump-section", f".hip_fatbin={fat}", str(obj)], check=True, capture_output=True)
    subprocess.run([str(LLVM / "clang-offload-bundler"), "--type=o", f"--targets={TARGET}", f"--input={fat}",
                    f"--output={co}", "--unbundle"], check=True, capture_output=True)
    notes = subprocess.run([str(LLVM / "llvm-readelf"), "--notes", str(co)], check=True, capture_output=True, text=True).stdout
    out, cur = [], None
    for line in notes.splitlines():
        m = re.match(r"\s*(?:- )?(\.[a-z_]+):\s*(.*)$", line)
        if not m:
            continue
        key, val = m.group(1), m.group(2).strip().strip("'")
        if key == ".agpr_count" and line.lstrip().startswith("- "):
            cur = {}
            out.append(cur)
        if cur is not None and key in FIELDS:
            cur[key] = val
    return [k for k in out if ".name" in k]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--all", action="store_true")
    ap.add_argument("--md", action="store_true")
    args = ap.parse_args()
    rows, bad = [], []
    with tempfile.TemporaryDirectory() as td:
        for obj in sorted(CSRC.glob("*.o")):
            ks = kernels_of(obj, Path(td))
            for k, nice in zip(ks, demangle([k[".name"] for k in ks])):
                r = dict(file=obj.stem, kernel=re.sub(r"^void ", "", nice).split("(")[0],
                         vgpr=int(k.get(".vgpr_count", 0)), agpr=int(k.get(".agpr_count", 0)), sgpr=int(k.get(".sgpr_count", 0)),
                         lds=int(k.get(".group_segment_fixed_size", 0)), scratch=int(k.get(".private_segment_fixed_size", 0)),
                         vspill=int(k.get(".vgpr_spill_count", 0)), sspill=int(k.get(".sgpr_spill_count", 0)),
                         wg=int(k.get(".max_flat_workgroup_size", 0)))
                rows.append(r)
                if r["scratch"] or r["vspill"]:
                    bad.append(r)
    show = rows if (args.all or args.md) else bad
    if args.md:
        print("| file | kernel | VGPR | AGPR | SGPR | LDS B | scratch B | max WG |\n|---|---|---|---|---|---|---|---|")
        for r in show:
            print(f"| {r['file']} | `{r['kernel']}` | {r['vgpr']} | {r['agpr']} | {r['sgpr']} | {r['lds']} | {r['scratch']} | {r['wg']} |")
    else:
        for r in show:
            print(f"{r['file']:14s} vgpr {r['vgpr']:3d} agpr {r['agpr']:3d} sgpr {r['sgpr']:3d} lds {r['lds']:6d} scratch {r['scratch']:4d} "
                  f"vspill {r['vspill']:3d}  {r['kernel']}")
    print(f"{len(rows)} kernels, {len(bad)} with scratch or spills", file=sys.stderr)
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
