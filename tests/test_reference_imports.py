"""Zero-change route: the reference's own modules import against THIS repository's
`depth_diff_gaussian_rasterization_min` and `simple_knn` packages.  Runs only where /root/reference exists (the build
container); the GPU box skips it.  Nothing of the reference is executed on a device here."""
import importlib
import os
import sys
import types

import pytest

REF = "/root/reference"
pytestmark = pytest.mark.skipif(not os.path.isdir(REF), reason="reference checkout not present")


@pytest.fixture()
def reference_on_path():
    added = []
    saved = {k: sys.modules.get(k) for k in ("plyfile", "utils", "scene", "gaussian_renderer", "arguments")}
    if "plyfile" not in sys.modules:                       # not installed here; only PlyData/PlyElement names are needed
        m = types.ModuleType("plyfile")
        m.PlyData = m.PlyElement = object
        sys.modules["plyfile"] = m
        added.append("plyfile")
    # scene/__init__.py pulls in the dataset readers (imageio, cv2, ...): register an empty `scene` package that only
    # points at the directory, so that `scene.gaussian_model` is found without running it
    pkg = types.ModuleType("scene")
    pkg.__path__ = [os.path.join(REF, "scene")]
    sys.modules["scene"] = pkg
    sys.path.insert(0, REF)
    try:
        yield
    finally:
        sys.path.remove(REF)
        for k in list(sys.modules):
            if k.split(".")[0] in ("utils", "scene", "gaussian_renderer", "arguments") and getattr(sys.modules[k], "__file__", "") and \
                    str(sys.modules[k].__file__).startswith(REF):
                del sys.modules[k]
        sys.modules.pop("scene", None)
        for k in added:
            sys.modules.pop(k, None)
        for k, v in saved.items():
            if v is not None:
                sys.modules[k] = v


def test_reference_renderer_and_model_import_against_our_packages(reference_on_path):
    import depth_diff_gaussian_rasterization_min as ours
    import simple_knn._C as knn
    gm = importlib.import_module("scene.gaussian_model")               # imports simple_knn._C.distCUDA2 (:19)
    assert gm.distCUDA2 is knn.distCUDA2
    gr = importlib.import_module("gaussian_renderer")                  # imports the rasterizer package (:14)
    assert gr.GaussianRasterizationSettings is ours.GaussianRasterizationSettings
    assert gr.GaussianRasterizer is ours.GaussianRasterizer
    assert callable(gr.render)
    model = gm.GaussianModel(3)                                        # constructor allocates nothing on a device
    assert model.max_sh_degree == 3 and model.get_xyz.numel() == 0
    # the settings tuple the reference's render() builds (gaussian_renderer/__init__.py:37-50) has our field order
    fields = ours.GaussianRasterizationSettings._fields
    assert fields == ("image_height", "image_width", "tanfovx", "tanfovy", "bg", "scale_modifier", "viewmatrix",
                      "projmatrix", "sh_degree", "campos", "prefiltered", "debug")


def test_densify_patch_installs_on_the_reference_class(reference_on_path):
    from luciddreamer_amd import densify
    gm = importlib.import_module("scene.gaussian_model")
    original = gm.GaussianModel.prune_points
    try:
        densify.patch(gm.GaussianModel)
        for name in ("prune_points", "densification_postfix", "densify_and_clone", "densify_and_split",
                     "densify_and_prune", "save_ply", "load_ply"):
            assert getattr(gm.GaussianModel, name).__module__ in ("luciddreamer_amd.densify",), name
    finally:
        gm.GaussianModel.prune_points = original
