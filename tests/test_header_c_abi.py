"""include/recon_hip.h must be a plain C header (the drop-in boundary is a C ABI, usable from cgo / JNI / ctypes)."""
import os
import shutil
import subprocess

import pytest

from conftest import ROOT


@pytest.mark.skipif(shutil.which("gcc") is None, reason="gcc not available")
def test_header_compiles_as_c(tmp_path):
    src = tmp_path / "t.c"
    src.write_text('#include "recon_hip.h"\n'
                   'int use(void) { recon_graph g; recon_gat_atp_args a; recon_prop_args p; recon_gcn_args c;\n'
                   '  (void)g; (void)a; (void)p; (void)c; return RECON_OK + RECON_ATP_BWD_ALL + RECON_ACT_RELU; }\n')
    subprocess.check_call(["gcc", "-std=c99", "-Wall", "-Werror", "-pedantic", "-fsyntax-only",
                           "-I", os.path.join(ROOT, "include"), str(src)])
