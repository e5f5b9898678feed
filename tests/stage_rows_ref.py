"""Test helper: global row of every native g staging slot of knot k, derived from the layout through the
host emulation (the product derives the same table in hipnlp_stage_rows)."""
import ctypes as C

import numpy as np


def stage_rows_from_blocks(emu, horizon):
    emu.lib.hostemu_stage_rows.argtypes = [C.c_void_p, C.c_int, C.POINTER(C.c_int)]
    out = []
    for k in range(horizon):
        rows = np.zeros(550, np.int32)
        emu.lib.hostemu_stage_rows(C.c_void_p(emu.h), k, rows.ctypes.data_as(C.POINTER(C.c_int)))
        out.append(rows)
    return out
