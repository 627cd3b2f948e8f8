"""The HIP path against the vectors of tools/pin_upstream.py (tinycudann / nerfacc / torch_efficient_distloss on seeded inputs).
With tests/golden/upstream_*.npz committed by a maintainer who holds the packages these tests are the pin of the third-party
arithmetic on the product path; until then they run on vectors the same script produces over the oracle stand-in, so that
format, seeded-input rules and checkers are exercised on every GPU run."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from tests.test_cpu_oracle import _upstream_files  # noqa: E402


@pytest.fixture(scope='module')
def vectors(tmp_path_factory):
    assert torch.cuda.is_available()
    paths, pinned = _upstream_files(tmp_path_factory.mktemp('upstream'))
    return {k: np.load(p, allow_pickle=False) for k, p in paths.items()}, pinned


@pytest.mark.parametrize('dtype', ['fp16', 'bf16'])
def test_fields_against_upstream_vectors(vectors, dtype):
    """tcnn.NetworkWithInputEncoding (PeRF's density and colour nets, ngp_nerf.py:96-134) forward + parameter gradient, and
    tcnn.Encoding with Smoothstep incl. the double backward of SphereDistanceField (pano_joint_predictor.py:30-67)."""
    from tests import upstream_check as C
    print(C.hip_vs_tcnn(vectors[0]['tcnn'], dtype))


def test_marching_and_compositing_against_upstream_vectors(vectors):
    """OccGridEstimator.sampling (traverse_grids + early termination), render_weight_from_density, accumulate_along_rays
    (nerf_renderer.py:145-183): ray_indices / t_starts / t_ends bit for bit, weights to fp32 rounding."""
    from tests import upstream_check as C
    print(C.hip_vs_nerfacc(vectors[0]['nerfacc']))


def test_distloss_against_upstream_vectors(vectors):
    from tests import upstream_check as C
    print(C.hip_vs_distloss(vectors[0]['distloss']))
