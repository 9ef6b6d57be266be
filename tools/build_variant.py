"""Build the specialised step kernel of a level with extra compiler flags into tools/ab/<name>.hsaco (A/B runs with
MJRL_SPEC_OBJECT).  Usage: build_variant.py name [level.xml] -- flags..."""
import os, subprocess, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as entry
entry.load_package()
from mjrl_amd import blob, kernel_cache, levels, mjcf
args = sys.argv[1:]
extra = args[args.index("--") + 1:] if "--" in args else []
args = args[:args.index("--")] if "--" in args else args
name = args[0]
level = args[1] if len(args) > 1 else "two_agent.xml"
sizes = kernel_cache.blob_sizes(blob.pack(mjcf.compile_mjcf(levels.level_path(level))))
os.makedirs(os.path.join(ROOT, "tools", "ab"), exist_ok=True)
out = os.path.join(ROOT, "tools", "ab", name + ".hsaco")
with tempfile.TemporaryDirectory() as tmp:
    hdr = os.path.join(tmp, "spec.h")
    open(hdr, "w").write(kernel_cache.spec_header(sizes))
    cmd = [kernel_cache.hipcc(), "--genco", *kernel_cache.FLAGS, *extra, f'-DMJRL_SPEC_HEADER="{hdr}"', "-I", kernel_cache.CSRC,
           os.path.join(kernel_cache.CSRC, "mjrl_spec_kernel.hip"), "-o", out]
    subprocess.run(cmd, check=True)
print(out)
