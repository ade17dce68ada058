"""CPU checks of the .ply reader and of the attribute layout (luciddreamer_amd.densify):
the reader is pure numpy; the writer's device part is covered by tests/test_gpu_densify.py."""
import os

import numpy as np
import pytest
import torch

from luciddreamer_amd import densify as D
from oracle import densify_oracle as O


def test_attribute_names_match_reference_order():
    # construct_list_of_attributes (gaussian_model.py:176-191): xyz, normals, f_dc_*, f_rest_*, opacity, scale_*, rot_*
    names = D.ply_attribute_names(15)
    assert names[:6] == ["x", "y", "z", "nx", "ny", "nz"]
    assert names[6:9] == ["f_dc_0", "f_dc_1", "f_dc_2"]
    assert names[9] == "f_rest_0" and names[53] == "f_rest_44" and names[54] == "opacity"
    assert names[55:58] == ["scale_0", "scale_1", "scale_2"] and names[58:] == ["rot_0", "rot_1", "rot_2", "rot_3"]
    assert len(names) == 62


@pytest.mark.parametrize("fmt", ["binary_little_endian", "ascii"])
def test_read_ply_roundtrip(tmp_path, fmt):
    g = torch.Generator().manual_seed(0)
    P = 37
    m = {"params": {"xyz": torch.randn(P, 3, generator=g), "f_dc": torch.randn(P, 1, 3, generator=g),
                    "f_rest": torch.randn(P, 15, 3, generator=g), "opacity": torch.randn(P, 1, generator=g),
                    "scaling": torch.randn(P, 3, generator=g), "rotation": torch.randn(P, 4, generator=g)}}
    rows = O.ply_rows(m).numpy().astype(np.float32)
    names = D.ply_attribute_names(15)
    path = os.path.join(tmp_path, "pc.ply")
    header = f"ply\nformat {fmt} 1.0\ncomment test\nelement vertex {P}\n" + "".join(f"property float {n}\n" for n in names) + "end_header\n"
    with open(path, "wb") as f:
        f.write(header.encode())
        if fmt == "ascii":
            for r in rows:
                f.write((" ".join(repr(float(v)) for v in r) + "\n").encode())
        else:
            f.write(rows.astype("<f4").tobytes())
    v = D.read_ply(path)
    assert list(v.keys()) == names
    got = np.stack([v[k] for k in names], axis=1)
    assert np.array_equal(got, rows)
    # f_rest is stored channel-major: f_rest_{c*15+k} == features_rest[:, k, c]
    assert np.array_equal(v["f_rest_16"], m["params"]["f_rest"][:, 1, 1].numpy())


def test_read_ply_rejects_other_files(tmp_path):
    p = os.path.join(tmp_path, "x.ply")
    open(p, "wb").write(b"not a ply\n")
    with pytest.raises(RuntimeError):
        D.read_ply(p)
    open(p, "wb").write(b"ply\nformat binary_little_endian 1.0\nelement vertex 1\nproperty double x\nend_header\n" + b"\0" * 8)
    with pytest.raises(RuntimeError):
        D.read_ply(p)


def test_densify_refuses_cpu_tensors():
    with pytest.raises(RuntimeError):
        D.RowStore({"xyz": torch.zeros(4, 3)})
