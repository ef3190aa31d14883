"""Scratch bytes and in-iteration-loop scratch / MFMA / LDS-DMA instruction counts of every N = 30 instantiation of admm_tile16_kernel
(the numbers tests/test_isa.py pins).      python tools/t16_isa_stats.py"""
import re, sys, importlib.util
from pathlib import Path
ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))
import accelerated_tinympc_amd as T
spec = importlib.util.spec_from_file_location('ti', ROOT / 'tests/test_isa.py'); ti = importlib.util.module_from_spec(spec); spec.loader.exec_module(ti)
for src in ("admm_tile16.hip", "admm_tile16_pi.hip"):
    txt = T.build.device_asm(src).read_text()
    sizes = ti.scratch_sizes(txt)
    for name, lines in ti.kernels_of(txt).items():
        if "admm_tile16_kernelILi30E" not in name: continue
        a, b = ti.iteration_loop(lines)
        cnt = lambda p: sum(1 for l in lines[a:b] if l.startswith(p))
        print(src, name[len('_ZN7tinympc18admm_tile16_kernelI'):-len('EvNS_9RowParamsE')], 'scratch', sizes[name], 'in-loop scratch', cnt("scratch_"), 'mfma', cnt("v_mfma"),
              'len', b - a, 'glds', cnt("global_load_lds"))
