"""Dev: what VALU work between fp32 MFMAs costs the MFMA pipe (ps_diag_mfma_mix)."""
import os; os.environ.setdefault("PS_DEV_ENV", "1")   # developer switches (PS_*) are read only under PS_DEV_ENV=1
import os, sys, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from precondition_amd._lib import lib
torch.zeros(1, device="cuda:0")
L = lib()
for wgs in (2, 1):
  for wide in (0, 1):
    row = []
    for nv in (0, 4, 8, 16, 32, 64):
      v = C.c_double(0)
      rc = L.ps_diag_mfma_mix(None, nv, wide, wgs, C.byref(v))
      row.append("%d:%.1f" % (nv, v.value) if rc == 0 else "%d:rc%d" % (nv, rc))
    print("wgs/CU %d  %s adds per 16 MFMAs -> TFLOP/s:" % (wgs, "64-bit" if wide else "32-bit"), "  ".join(row), flush=True)
