"""CPU: the C-ABI library builds for gfx950, loads, and exports every symbol include/ggpm_hip.h declares."""
import os
import re

from ggpm_amd import _lib, build

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared_functions():
    text = open(os.path.join(ROOT, "include", "ggpm_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(ggpm_[a-z0-9_]+)\s*\(", text)))


def test_header_and_binding_agree():
    assert _declared_functions() == sorted(_lib.SIGNATURES.keys())


def test_library_builds_and_exports_every_symbol():
    build.build(verbose=False)
    lib = _lib.load()
    for name in _declared_functions():
        assert hasattr(lib, name), name
    # host-only entry points can be called without a GPU
    assert lib.ggpm_version() >= 100
    assert lib.ggpm_padded_hidden(300) == 304 and lib.ggpm_padded_hidden(16) == 16
    assert lib.ggpm_error_string(1).decode().startswith("invalid")
    # three matrices in the largest packed form (three bf16 planes, 320 padded columns: 48 * Hp * ceil(Hp / 32) floats) + bias
    assert lib.ggpm_gru_pack_floats(300) == 3 * 48 * 304 * 10 + 304
    assert lib.ggpm_gemm_workspace_bytes(300, 300, 100) == 0


def test_ctypes_structs_match_the_header_layout(tmp_path):
    """The structs that cross the C ABI (ggpm_enc_dims, ggpm_decode_steps, ggpm_sched_in): a C program compiled against
    include/ggpm_hip.h prints sizeof and every field offset; the ctypes mirrors must agree."""
    import ctypes
    import subprocess
    import pytest
    import torch
    if torch.cuda.is_initialized():
        # this is a CPU test ("-m 'not gpu'" never initialises HIP): under an unusual selection that has, skip rather than
        # fork a GPU-initialised process for a compiler run (starting fresh children is allowed on the pool -- replacing the
        # GPU process is what is refused -- but a fork of a process with live HIP state buys nothing here)
        pytest.skip("the ABI probe runs gcc in child processes: kept to processes that have not initialised the GPU")
    from ggpm_amd.atom_decode import DecodeSteps
    from ggpm_amd.fused import EncDims
    from ggpm_amd.schedule_native import SchedIn
    from ggpm_amd.tree_decode import TreeLevelC, TreeLevelGrads, TreeLevelViews
    structs = {"ggpm_enc_dims": EncDims, "ggpm_decode_steps": DecodeSteps, "ggpm_sched_in": SchedIn,
               "ggpm_tree_level": TreeLevelC, "ggpm_tree_level_views": TreeLevelViews, "ggpm_tree_level_grads": TreeLevelGrads}
    lines = []
    for cname, cls in structs.items():
        lines.append('printf("%s %%zu", sizeof(%s));' % (cname, cname))
        for f, _ in cls._fields_:
            lines.append('printf(" %%zu", offsetof(%s, %s));' % (cname, f))
        lines.append('printf("\\n");')
    src = tmp_path / "abi.c"
    src.write_text('#include <stdio.h>\n#include <stddef.h>\n#include "ggpm_hip.h"\nint main(void) {\n%s\nreturn 0; }\n'
                   % "\n".join(lines))
    exe = tmp_path / "abi"
    subprocess.check_call(["gcc", "-std=c11", "-I", os.path.join(ROOT, "include"), str(src), "-o", str(exe)])
    out = subprocess.check_output([str(exe)]).decode().split("\n")
    for line in filter(None, out):
        name, *nums = line.split()
        cls = structs[name]
        want = [ctypes.sizeof(cls)] + [getattr(cls, f).offset for f, _ in cls._fields_]
        assert [int(x) for x in nums] == want, (name, nums, want)


def test_integration_md_struct_stubs_match_the_bindings():
    """The ctypes stubs a maintainer would copy out of INTEGRATION.md must declare the structs exactly as the
    bindings in ggpm_amd/ do (field names, order, C types): a short struct makes the driver read past its end."""
    import ctypes
    from ggpm_amd.fused import EncDims
    text = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    blocks = re.findall(r"```python\n(.*?)```", text, flags=re.S)
    found = {}
    for b in blocks:
        for m in re.finditer(r"^class (\w+)\(ctypes\.Structure\):.*?\n((?:[ \t]+.*\n)+)", b, flags=re.M):
            ns = {"ctypes": ctypes}
            exec(m.group(0), ns)
            found[m.group(1)] = ns[m.group(1)]
    assert "EncDims" in found
    doc, real = found["EncDims"], EncDims
    assert [(n, t) for n, t in doc._fields_] == [(n, t) for n, t in real._fields_]
    assert ctypes.sizeof(doc) == ctypes.sizeof(real)
    # the constructor call shown next to it passes one value per field
    call = re.search(r"dims = EncDims\((.*?)\)\s*#", text, flags=re.S).group(1)
    n_args = len([a for a in call.replace("\n", " ").split(",") if a.strip()])
    n_star = call.count("*graph_shapes") * 3 + call.count("*tree_shapes") * 4       # 4 + 5 values behind the two stars
    assert n_args + n_star == len(real._fields_), (n_args, n_star, len(real._fields_))


def test_product_path_refuses_cpu_tensors():
    """No CPU fallback: the wrappers raise instead of silently computing elsewhere."""
    import pytest
    import torch
    from ggpm_amd import functional as F_
    with pytest.raises(RuntimeError):
        F_.csr_from_padded(torch.zeros(3, 2, dtype=torch.int64), ncols=3)
    with pytest.raises(RuntimeError):
        F_.linear([torch.zeros(2, 4)], [4], torch.zeros(3, 4), None)
