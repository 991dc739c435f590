"""VOS / RefVOS driver (univs_amd/inference/video_vos.py) with its bookkeeping on the GPU, against the
REFERENCE's InferenceVideoVOS on the scripted scene (tests/golden/g15*): per-clip `targets[0]` states and the id maps /
per-expression masks the reference wrote as PNG files."""
import os

import numpy as np
import pytest
import torch

from tests import cases
from tests.test_clip_loop_cpu import compare_vos_states, run_vos

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("name,mode", [("g15a_vos_prompt", "prompt"), ("g15b_vos_prompt_learn", "prompt+learn"),
                                       ("g15d_vos_learn", "learn")])
def test_vos_driver_on_device_matches_reference(cuda, golden_dir, name, mode):
    from univs_amd.inference.video_vos import FrameAnnotations
    g = np.load(os.path.join(golden_dir, name + ".npz"))
    calls, states, results = run_vos(mode, cases.vos_targets_sot(FrameAnnotations), device=cuda)
    compare_vos_states(calls, states, g)
    idmaps = torch.cat(results)
    assert idmaps.dtype == torch.uint8
    assert torch.equal(idmaps, torch.from_numpy(g["result_idmaps"]))
    assert set(idmaps.unique().tolist()) == {0, *cases.VOS_OBJECTS}


def test_rvos_driver_on_device_matches_reference(cuda, golden_dir):
    g = np.load(os.path.join(golden_dir, "g15c_rvos_grounding.npz"))
    targets = cases.vos_targets_grounding()
    calls, states, results = run_vos("prompt", targets, device=cuda)
    compare_vos_states(calls, states, g)
    assert all(r["ids"] == targets[0]["exp_obj_ids"] for r in results)
    per_exp = torch.cat([r["masks"] for r in results], dim=1)
    for n, eid in enumerate(targets[0]["exp_obj_ids"]):
        assert torch.equal(per_exp[n], torch.from_numpy(g[f"result_exp{eid}"])), eid


def test_viposeg_panoptic_vos_on_device_matches_reference(cuda, golden_dir):
    from univs_amd.inference.video_vos import FrameAnnotations
    g = np.load(os.path.join(golden_dir, "g15h_viposeg_prompt_learn.npz"))
    case = cases.VIPOSEG_CASE
    targets = cases.vos_targets_sot(FrameAnnotations, case, objects=cases.VIPOSEG_OBJECTS,
                                    class_offset=cases.VIPOSEG_CLASS_START, dataset="viposeg_val")
    calls, states, results = run_vos("prompt+learn", targets, case=case, objects=cases.VIPOSEG_OBJECTS,
                                     stuff_dataset_ids=cases.VIPOSEG_STUFF_IDS, device=cuda)
    compare_vos_states(calls, states, g)
    assert torch.equal(torch.cat(results), torch.from_numpy(g["result_idmaps"]))
