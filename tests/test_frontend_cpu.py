"""the logic of the reference's Gradio page (/root/reference/latentblending/gradio_ui.py:29-262 -
``MultiUserRouter`` / ``BlendingVariableHolder``) without the widgets: preview renders, the edited list of key frames, the
JSON written after every edit, and ``generate_movie`` = the swap_forward / recycle_img1 chain over that list."""
import json
import os

import numpy as np
import pytest

from oracle import pipe as OP
from oracle import sdxl_ref as R


@pytest.fixture()
def cpu_backend():
    from latentblending_amd.backend import set_backend
    set_backend(R.TorchCpuBackend())
    yield
    set_backend(None)


def engine():
    from latentblending_amd import BlendingEngine
    np.random.seed(0)
    p = OP.StableDiffusionXLPipeline(turbo=True, unet_cfg=R.tiny_unet_cfg(), vae_cfg=R.tiny_vae_cfg())
    be = BlendingEngine(p, metric=R.OracleLPIPS(7), verbose=False)
    be.set_num_inference_steps(2)
    be.set_branching(depth_strength=0.5, nmb_max_branches=2)
    return be


class Select:                       # what gr.SelectData carries that the callbacks read
    def __init__(self, index):
        self.index = index


def test_router_drives_the_page_logic_end_to_end(tmp_path, cpu_backend):
    from latentblending_amd import MultiUserRouter, replay
    from latentblending_amd.movie import read_movie_header
    shared = engine()
    mur = MultiUserRouter(engines={"tiny-turbo": shared}, dp_out=str(tmp_path))
    uid = mur.register_new_user("tiny-turbo", 64, 64)
    other = mur.register_new_user("tiny-turbo", 128, 128)          # a second user of the same engine, another size
    holder = mur.user_blendingvariableholder[uid]
    holder.nmb_preview_images = 2
    assert mur.add_image_to_video(uid) == []                        # no prompt yet: refused, as upstream
    picked = []
    for k, prompt in enumerate(["a reef", "an alien planet", "fog over a harbour"]):
        previews = mur.compute_imgs(uid, prompt, "blurry")
        assert len(previews) == 2 and all(os.path.isfile(f) for f in previews) and len(set(holder.list_seeds)) == 2
        assert mur.add_image_to_video(uid) == mur.get_list_images_movie(uid)         # nothing selected: refused
        mur.preview_img_selected(uid, Select(k % 2), None)
        shown = mur.add_image_to_video(uid)
        picked.append((prompt, holder.list_seeds[k % 2]))
        assert shown[-1] == previews[k % 2] and len(shown) == k + 1
    # the JSON the page saves after every edit: header first (size of THIS user, not of the other one), then the items
    saved = json.load(open(holder.fp_json))
    assert saved[0] == {"settings": "sdxl", "width": 64, "height": 64, "num_inference_steps": 2}
    assert [(it["prompt"], it["seed"]) for it in saved[1:]] == picked and [it["iteration"] for it in saved[1:]] == [0, 1, 2]
    # list edits: later / earlier swap neighbours, the ends are refused, delete removes
    mur.movie_img_selected(uid, Select(0), None)
    mur.img_movie_later(uid)
    assert [it["prompt"] for it in holder.data] == ["an alien planet", "a reef", "fog over a harbour"]
    mur.movie_img_selected(uid, Select(2), None)
    mur.img_movie_later(uid)                                        # last image: refused (the reference raises IndexError here)
    assert [it["prompt"] for it in holder.data] == ["an alien planet", "a reef", "fog over a harbour"]
    mur.movie_img_selected(uid, Select(1), None)
    mur.img_movie_earlier(uid)
    assert [it["prompt"] for it in holder.data] == ["a reef", "an alien planet", "fog over a harbour"]
    mur.movie_img_selected(uid, Select(0), None)
    mur.img_movie_earlier(uid)                                      # first image: refused
    mur.movie_img_selected(uid, Select(3), None)
    mur.img_movie_delete(uid)                                       # one past the end: refused
    assert len(holder.data) == 3
    # generate_movie = the chain the scripts run (example_multi_trans_json.py:24-71) over the edited list
    fp = mur.generate_movie(uid, 0.2)
    assert fp == holder.fp_movie and os.path.isfile(fp) and read_movie_header(fp)[3] > 0
    assert all(os.path.isfile(os.path.join(str(tmp_path), f"tmp_part_00{i}.mp4")) for i in range(2))
    # the same list replayed from the saved JSON by the script-level driver gives the same frames
    holder.write_json()
    got = []
    with holder.session.bound() as be:
        chain = replay.run_movie_json(be, holder.fp_json, None)
        got = [[np.asarray(f).copy() for f in seg] for seg in chain]
    assert len(got) == 2 and all(len(seg) >= 3 for seg in got)
    # users never see each other: the other user's session still has its own size and no prompts of this one
    with mur._sessions.session(other).bound() as be:
        assert (be.dh.width_img, be.dh.height_img) == (128, 128) and be.prompt1 == ""
    assert shared.prompt1 == "" and getattr(shared, "_bound_session", None) is None
    mur.movie_img_selected(uid, Select(1), None)
    mur.img_movie_delete(uid)
    assert [it["prompt"] for it in holder.data] == ["a reef", "fog over a harbour"]


def test_launch_ui_needs_gradio(cpu_backend):
    from latentblending_amd.frontend import MultiUserRouter, launch_ui
    try:
        import gradio  # noqa: F401
        pytest.skip("gradio is installed: the page itself is exercised by hand")
    except ImportError:
        pass
    with pytest.raises(ImportError, match="gradio"):
        launch_ui(MultiUserRouter(engines={"m": engine()}), launch=False)
