"""Registers, scratch and LDS of a level's specialised step kernel as hipcc builds it now (no GPU needed).
Usage: spec_resources.py [level.xml] [-- extra flags]"""
import os, re, subprocess, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as entry
entry.load_package()
from mjrl_amd import blob, kernel_cache, levels, mjcf
args = sys.argv[1:]
extra = args[args.index("--") + 1:] if "--" in args else []
args = args[:args.index("--")] if "--" in args else args
level = args[0] if args else "two_agent.xml"
sizes = kernel_cache.blob_sizes(blob.pack(mjcf.compile_mjcf(levels.level_path(level))))
LLVM = "/opt/rocm/lib/llvm/bin"
with tempfile.TemporaryDirectory() as tmp:
    hdr = os.path.join(tmp, "spec.h")
    open(hdr, "w").write(kernel_cache.spec_header(sizes))
    out = os.path.join(tmp, "k.hsaco")
    subprocess.run([kernel_cache.hipcc(), "--genco", *kernel_cache.FLAGS, *extra, f'-DMJRL_SPEC_HEADER="{hdr}"', "-I",
                    kernel_cache.CSRC, os.path.join(kernel_cache.CSRC, "mjrl_spec_kernel.hip"), "-o", out], check=True)
    co = os.path.join(tmp, "k.co")
    subprocess.run([f"{LLVM}/clang-offload-bundler", "--unbundle", "--type=o", "--targets=hipv4-amdgcn-amd-amdhsa--gfx950",
                    f"--input={out}", f"--output={co}"], check=True)
    notes = subprocess.run([f"{LLVM}/llvm-readelf", "--notes", co], capture_output=True, text=True).stdout
    for key in (".vgpr_count", ".agpr_count", ".sgpr_count", ".private_segment_fixed_size", ".vgpr_spill_count", ".sgpr_spill_count"):
        print(key, re.findall(re.escape(key) + r":\s*(\d+)", notes))
    if os.environ.get("KEEP_ASM"):
        asm = subprocess.run([f"{LLVM}/llvm-objdump", "-d", co], capture_output=True, text=True).stdout
        open(os.environ["KEEP_ASM"], "w").write(asm)
        print("asm lines", asm.count("\n"), "global_load", asm.count("global_load"), "vmcnt waits", asm.count("s_waitcnt vmcnt"))
