"""Code-object audit of libilqr_hip.so: registers, scratch (private segment), LDS and spills of every gfx950 kernel, from the metadata notes of the
fat binary's device code object.  `python scripts/audit_code_objects.py [--all]` prints the kernels that carry a private segment or sit at the
512-VGPR ceiling (all kernels with --all); the table of DESIGN.md section 5.5 is this script's output.  Needs only the ROCm LLVM tools."""
import os, re, subprocess, sys, tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LLVM = "/opt/rocm/lib/llvm/bin"
lib = os.path.join(ROOT, "ilqr_planner_amd", "libilqr_hip.so")
notes = ""
with tempfile.TemporaryDirectory() as td:
    fat = os.path.join(td, "fat.bin")
    subprocess.check_call([f"{LLVM}/llvm-objcopy", "--dump-section", f".hip_fatbin={fat}", lib, os.path.join(td, "unused.so")])
    blob = open(fat, "rb").read()
    magic = b"__CLANG_OFFLOAD_BUNDLE__"
    starts = [m.start() for m in re.finditer(re.escape(magic), blob)]
    for n, a in enumerate(starts):  # one bundle per translation unit
        part = os.path.join(td, f"b{n}.bin")
        open(part, "wb").write(blob[a:(starts[n + 1] if n + 1 < len(starts) else len(blob))])
        tgt = subprocess.run([f"{LLVM}/clang-offload-bundler", "--list", "--type=o", f"--input={part}"], capture_output=True, text=True).stdout.split()
        dev = [t for t in tgt if "gfx950" in t]
        if not dev:
            continue
        co = os.path.join(td, f"dev{n}.co")
        subprocess.check_call([f"{LLVM}/clang-offload-bundler", "--unbundle", "--type=o", f"--targets={dev[0]}", f"--input={part}", f"--output={co}"])
        notes += subprocess.run([f"{LLVM}/llvm-readelf", "--notes", co], capture_output=True, text=True).stdout
kern = []
for blk in notes.split("- .agpr_count:")[1:]:
    g = lambda k: (re.search(r"\.%s:\s+(\S+)" % k, blk) or [None, "0"])[1]
    name = subprocess.run(["c++filt", g("name")], capture_output=True, text=True).stdout.strip()
    kern.append(dict(name=name, vgpr=int(g("vgpr_count")), sgpr=int(g("sgpr_count")), scratch=int(g("private_segment_fixed_size")), lds=int(g("group_segment_fixed_size")),
                     vspill=int(g("vgpr_spill_count")), sspill=int(g("sgpr_spill_count"))))
show_all = "--all" in sys.argv
sel = [k for k in kern if show_all or k["scratch"] > 0 or k["vgpr"] >= 512]
print(f"{len(kern)} kernels; {sum(1 for k in kern if k['scratch'] > 0)} with a private segment, {sum(1 for k in kern if k['vgpr'] >= 512)} at 512 VGPRs")
for k in sorted(sel, key=lambda k: (-k["scratch"], -k["vgpr"])):
    nm = re.sub(r"^void ilqr::", "", k["name"])
    nm = re.sub(r"\(ilqr::Bufs.*$|\(.*$", "", nm)
    print(f"{nm[:78]:78s} vgpr {k['vgpr']:3d} sgpr {k['sgpr']:3d} scratch {k['scratch']:5d} B  lds {k['lds']:6d} B  spills v{k['vspill']} s{k['sspill']}")
