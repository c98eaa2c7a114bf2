python - <<'PY' 2>&1 | grep -v amdgpu.ids
import sys, ctypes
sys.path.insert(0,'tools'); sys.path.insert(0,'.')
import conv_bench as cb
from audiodec_amd import native
lib = native.lib()
def flags():
    f = ctypes.c_int32(0); lib.adk_debug_flags(ctypes.byref(f)); return f.value
for sh in ("up0","up1x"):
    if sh == "up1x":
        cb.SHAPES["up1x"] = (256, 640, 1, 2, 1, 1, 5, 5, 2)
    for impl in (6, 2):
        for cfg in (-1, 0, 3, 4):
            for B in (256, 64):
                us, tf = cb.run(sh, cfg, B, 20, impl)
                print(sh, "impl", impl, "cfg", cfg, "B", B, f"{us:9.1f} us", "flags", flags(), flush=True)
PY
