import os, sys
sys.path.insert(0, os.getcwd())
import __graft_entry__ as entry
entry.load_package()
from mjrl_amd import mjcf, levels, blob, _capi
import torch
p = torch.cuda.get_device_properties(0)
print(p.name, "CUs", p.multi_processor_count, "shared/block", getattr(p, "shared_memory_per_block", None), "shared/mp", getattr(p, "shared_memory_per_multiprocessor", None), "regs/mp", getattr(p, "regs_per_multiprocessor", None))
for nconmax, njmax in [(16, 80), (6, 40), (2, 24)]:
    m = mjcf.compile_mjcf(levels.level_path("two_agent.xml"), nconmax=nconmax, njmax=njmax)
    h = _capi.Handle(blob.pack(m), 64)
    print(njmax, "lds KiB", h.size("lds_doubles") * 8 / 1024, "blocks per CU (runtime):", h.size("blocks_per_cu"))
