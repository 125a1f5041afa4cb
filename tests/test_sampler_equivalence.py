"""The estimator the kernels implement (counter-based sampling) equals the reference's
(StratifiedSampler + engine draws) in expectation and variance: protocol and tolerance in
tests/estimator.py.  CPU: oracle mode 0 against oracle mode 1.  GPU: the HIP image against mode 1."""
import numpy as np
import pytest

from raytracer_amd.host import DeviceScene
from tests import estimator as est
from tests import orc


@pytest.mark.parametrize("name", sorted(est.SCENES))
def test_counter_sampler_matches_stratified_sampler_cpu(built, name):
    make, w, h = est.SCENES[name]
    flat = make().initKdTree().flatten(1)
    ref, base = est.reference_side(flat, w, h)
    spp = est.N_GRID[0] * est.N_GRID[1]
    test = [orc.oracle_render(flat.desc, est.candidate_params(w, h, k), 0, threads=8)[0] / spp for k in range(est.SEEDS)]
    v = est.verdict(ref, base, test)
    print(name, v)
    assert v["ok"], v


@pytest.mark.gpu
@pytest.mark.parametrize("name", sorted(est.SCENES))
def test_hip_image_matches_stratified_sampler(built, name):
    make, w, h = est.SCENES[name]
    flat = make().initKdTree().flatten(1)
    ref, base = est.reference_side(flat, w, h, threads=16)
    spp = est.N_GRID[0] * est.N_GRID[1]
    dev = DeviceScene(flat)
    test = []
    for k in range(est.SEEDS):
        img = np.zeros((h, w, 3), np.float32)
        dev.render(est.candidate_params(w, h, k), img)
        test.append(img / spp)
    dev.close()
    v = est.verdict(ref, base, test)
    print(name, v)
    assert v["ok"], v
