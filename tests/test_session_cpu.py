"""SURVEY.md §8f rank 4 (minimal form): the persisted engine state round-trips through the reference's yml files, and
several users share ONE engine (weights, programs) through per-user sessions without seeing each other's state
(/root/reference/latentblending/blending_engine.py:709-728, gradio_ui.py:29-54)."""
import threading

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


def tiny_pipe(turbo):
    return OP.StableDiffusionXLPipeline(turbo=turbo, unet_cfg=R.tiny_unet_cfg(), vae_cfg=R.tiny_vae_cfg())


def frames_of(imgs):
    return [np.asarray(i).copy() for i in imgs]


def test_state_dict_round_trips_through_yml(tmp_path, cpu_backend):
    from latentblending_amd import BlendingEngine, yml_load, yml_save
    p = tiny_pipe(False)
    np.random.seed(0)
    be = BlendingEngine(p, metric=R.OracleLPIPS(7), verbose=False)
    be.set_dimensions((64, 64))
    be.set_num_inference_steps(6)
    be.set_guidance_scale(3.0)
    be.set_negative_prompt("blurry, pale")
    be.set_branch1_crossfeed(0.3, 0.5, 0.5)
    be.set_branching(depth_strength=0.5, nmb_max_branches=5)
    be.set_prompt1("photo of a reef")
    be.set_prompt2("rendering of an alien planet")
    be.seed1, be.seed2 = 420, 421
    state = be.get_state_dict()
    for key in ("prompt1", "prompt2", "seed1", "seed2", "width", "height", "num_inference_steps", "guidance_scale",
                "guidance_scale_mid_damper", "negative_prompt", "branch1_crossfeed_power", "parental_crossfeed_decay"):
        assert key in state                    # (the reference's key list, blending_engine.py:711-715, with its typos fixed)
    fp = str(tmp_path / "state.yml")
    yml_save(fp, state)
    want = frames_of(be.run_transition())
    tree = (list(be.tree_fracts), list(be.tree_idx_injection))

    p2 = tiny_pipe(False)
    be2 = BlendingEngine(p2, metric=R.OracleLPIPS(7), verbose=False)
    be2.load_state_dict(yml_load(fp))
    assert be2.get_state_dict() == state
    got = frames_of(be2.run_transition())
    assert (list(be2.tree_fracts), list(be2.tree_idx_injection)) == tree
    assert len(got) == len(want) and all(np.array_equal(a, b) for a, b in zip(got, want))


def test_sessions_isolate_users_of_one_engine(cpu_backend):
    """Two users with different settings interleave calls on ONE shared engine; each gets exactly what a dedicated engine
    gives, including a chained second transition that recycles the user's OWN previous tree."""
    from latentblending_amd import BlendingEngine, SessionRouter

    def user_a(be):
        be.set_num_inference_steps(6)
        be.set_guidance_scale(3.0)              # (> 1: this user runs classifier-free guidance with a negative prompt)
        be.set_negative_prompt("blurry")
        be.set_parental_crossfeed(0.8, 0.5, 0.5)
        be.set_branching(depth_strength=0.5, nmb_max_branches=4)
        be.set_prompt1("a reef"); be.set_prompt2("an alien planet")

    def user_b(be):
        be.set_num_inference_steps(4)
        be.set_branching(depth_strength=0.25, nmb_max_branches=3)
        be.set_prompt1("fog"); be.set_prompt2("a harbour")

    def run(be, pipe, **kw):                    # (ancestral sampler: every transition starts the noise tape afresh)
        pipe.noise.reset()
        return frames_of(be.run_transition(**kw))

    def dedicated(setup, second_prompt):
        np.random.seed(0)
        p = tiny_pipe(True)
        be = BlendingEngine(p, metric=R.OracleLPIPS(7), verbose=False)
        be.set_dimensions((64, 64))
        setup(be)
        first = run(be, p, fixed_seeds=[5, 6])
        be.swap_forward(); be.set_prompt2(second_prompt)
        second = run(be, p, recycle_img1=True, fixed_seeds=[6, 7])
        return first, second, list(be.tree_fracts)

    want_a, want_b = dedicated(user_a, "a forest"), dedicated(user_b, "a desert")

    np.random.seed(0)
    ps = tiny_pipe(True)
    shared = BlendingEngine(ps, metric=R.OracleLPIPS(7), verbose=False)
    router = SessionRouter({"turbo": shared})
    ua, ub = router.register_new_user("turbo", 64, 64), router.register_new_user("turbo", 64, 64)
    sa, sb = router.session(ua), router.session(ub)
    with sa.bound() as be:
        user_a(be)
    with sb.bound() as be:
        user_b(be)
    with sa.bound() as be:
        a1 = run(be, ps, fixed_seeds=[5, 6])
    with sb.bound() as be:                                                 # B runs between A's two transitions
        b1 = run(be, ps, fixed_seeds=[5, 6])
    with sa.bound() as be:
        assert be.num_inference_steps == 6 and be.dh.num_inference_steps == 6 and be.guidance_scale_base == 3.0
        be.swap_forward(); be.set_prompt2("a forest")
        a2 = run(be, ps, recycle_img1=True, fixed_seeds=[6, 7])
        fr_a = list(be.tree_fracts)
    with sb.bound() as be:
        assert be.num_inference_steps == 4 and be.negative_prompt is None and be.guidance_scale_base == 0.0
        be.swap_forward(); be.set_prompt2("a desert")
        b2 = run(be, ps, recycle_img1=True, fixed_seeds=[6, 7])
        fr_b = list(be.tree_fracts)
    for got, want in (((a1, a2, fr_a), want_a), ((b1, b2, fr_b), want_b)):
        assert got[2] == want[2]
        for x, y in zip(got[0] + got[1], want[0] + want[1]):
            assert np.array_equal(x, y)
    assert sa.get_state_dict()["prompt2"] == "a forest" and sb.get_state_dict()["prompt2"] == "a desert"


def test_sessions_serialise_concurrent_callers(cpu_backend):
    """Two threads drive two users of one engine at once: calls serialise on the engine's lock, results are per user."""
    from latentblending_amd import BlendingEngine, SessionRouter
    np.random.seed(0)
    shared = BlendingEngine(tiny_pipe(True), metric=R.OracleLPIPS(7), verbose=False)
    router = SessionRouter({"turbo": shared})
    users = [router.register_new_user("turbo", 64, 64) for _ in range(2)]
    out, errors = {}, []

    def work(uid, steps, seeds):
        try:
            from latentblending_amd.backend import set_backend
            set_backend(R.TorchCpuBackend())
            s = router.session(uid)
            with s.bound() as be:
                be.set_num_inference_steps(steps)
                be.set_branching(depth_strength=0.5, nmb_max_branches=3)
                be.set_prompt1("a"); be.set_prompt2("b")
            out[uid] = (frames_of(s.run_transition(fixed_seeds=seeds)), steps)
        except Exception as exc:       # pragma: no cover
            errors.append(exc)

    threads = [threading.Thread(target=work, args=(users[0], 4, [1, 2])), threading.Thread(target=work, args=(users[1], 6, [3, 4]))]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    assert not errors, errors
    for uid in users:
        frames, steps = out[uid]
        with router.session(uid).bound() as be:
            assert be.num_inference_steps == steps and len(be.tree_latents[0]) == steps and len(frames) == len(be.tree_fracts)


def test_new_users_never_inherit_another_users_state(cpu_backend):
    """A user registered AFTER, or DURING, somebody else's bound() block starts from the router's defaults - not from the
    prompts / embeddings / preset frames the shared engine happens to hold (round-4 advisor finding); bound() puts the
    engine back; two different sessions do not nest."""
    from latentblending_amd import BlendingEngine, EngineSession, SessionRouter
    np.random.seed(0)
    shared = BlendingEngine(tiny_pipe(True), metric=R.OracleLPIPS(7), verbose=False)
    router = SessionRouter({"turbo": shared})
    pristine = shared.text_embedding1           # (whatever a fresh engine holds: the operator's defaults)

    def same_embedding(a, b):
        import torch
        if a is None or b is None:
            return a is b
        return len(a) == len(b) and all((x is None and y is None) or (x is not None and y is not None and torch.equal(x, y)) for x, y in zip(a, b))
    ua = router.register_new_user("turbo", 64, 64)
    with router.session(ua).bound() as be:
        be.set_negative_prompt("secret negative")
        be.set_prompt1("user A's private prompt"); be.set_prompt2("another private prompt")
        be.set_branching(depth_strength=0.5, nmb_max_branches=3)
        be._preset_anchor_frames = ["A's frame", None]
        during = {}

        def register_from_another_thread():     # (blocks on the engine lock until A's block ends)
            during["uid"] = router.register_new_user("turbo", 64, 64)
        t = threading.Thread(target=register_from_another_thread)
        t.start()
        with pytest.raises(RuntimeError):       # a second session inside A's block, same thread
            with router.session(ua).__class__(shared, router._locks["turbo"], defaults=router._defaults["turbo"]).bound():
                pass
        with pytest.raises(RuntimeError):       # a router-less session built while the engine is bound
            EngineSession(shared, router._locks["turbo"])
    t.join()
    # the shared engine holds nothing of user A
    assert shared.prompt1 == "" and same_embedding(shared.text_embedding1, pristine) and shared.negative_prompt is None and shared._preset_anchor_frames is None
    ub = router.register_new_user("turbo", 64, 64)
    for uid in (ub, during["uid"]):
        with router.session(uid).bound() as be:
            assert be.prompt1 == "" and be.prompt2 == "" and same_embedding(be.text_embedding1, pristine) and same_embedding(be.text_embedding2, shared.text_embedding2)
            assert be.negative_prompt is None and be._preset_anchor_frames is None and not be.tree_fracts
    with router.session(ua).bound() as be:      # and A still has its own
        assert be.prompt1 == "user A's private prompt" and be.negative_prompt == "secret negative" and be._preset_anchor_frames == ["A's frame", None]
        with router.session(ua).bound() as again:       # the SAME session may re-enter
            assert again is be


def test_load_state_dict_resets_a_stale_negative_prompt(cpu_backend):
    """A state saved WITHOUT a negative prompt, loaded into an engine that has one: the stale prompt must go (round-4 advisor
    finding) - same frames as loading into a fresh engine."""
    from latentblending_amd import BlendingEngine
    np.random.seed(0)
    src = BlendingEngine(tiny_pipe(False), metric=R.OracleLPIPS(7), verbose=False)
    src.set_dimensions((64, 64)); src.set_num_inference_steps(4); src.set_guidance_scale(3.0)
    src.set_branching(depth_strength=0.5, nmb_max_branches=3)
    src.set_prompt1("a reef"); src.set_prompt2("an alien planet")
    src.seed1, src.seed2 = 3, 4
    state = src.get_state_dict()
    assert state["negative_prompt"] is None
    want = frames_of(src.run_transition())
    dirty = BlendingEngine(tiny_pipe(False), metric=R.OracleLPIPS(7), verbose=False)
    dirty.set_negative_prompt("blurry, pale, something stale")
    dirty.load_state_dict(state)
    assert dirty.negative_prompt is None and dirty.dh.negative_prompt == ""
    got = frames_of(dirty.run_transition())
    assert len(got) == len(want) and all(np.array_equal(a, b) for a, b in zip(got, want))


def test_failed_install_leaves_the_shared_engine_untouched(cpu_backend):
    """Round-5 advice: an install that raises half way (here: a step count the holder rejects) must neither leave this user's
    prompts / fields on the shared engine nor keep the engine marked as bound."""
    from latentblending_amd import BlendingEngine, EngineSession
    np.random.seed(0)
    shared = BlendingEngine(tiny_pipe(True), metric=R.OracleLPIPS(7), verbose=False)
    shared.set_prompt1("operator prompt")
    s = EngineSession(shared)
    with s.bound() as be:
        be.set_prompt1("user prompt")
    assert shared.prompt1 == "operator prompt"
    good = s._holder["num_inference_steps"]
    s._holder["num_inference_steps"] = "not a step count"
    before = dict(prompt1=shared.prompt1, steps=shared.dh.num_inference_steps, guidance=shared.dh.guidance_scale)
    with pytest.raises(Exception):
        with s.bound():
            pass
    assert getattr(shared, "_bound_session", None) is None
    assert dict(prompt1=shared.prompt1, steps=shared.dh.num_inference_steps, guidance=shared.dh.guidance_scale) == before
    s._holder["num_inference_steps"] = good
    with s.bound() as be:                       # the session itself survived the failed attempt
        assert be.prompt1 == "user prompt"


def test_router_less_sessions_of_one_engine_share_one_lock(cpu_backend):
    """Two ``EngineSession(be)`` built without a router exclude each other: a second thread BLOCKS while the first is inside
    ``bound()`` (a lock per session let both install themselves, or raised the "same thread" error from another thread)."""
    from latentblending_amd import BlendingEngine, EngineSession
    np.random.seed(0)
    shared = BlendingEngine(tiny_pipe(True), metric=R.OracleLPIPS(7), verbose=False)
    a, b = EngineSession(shared), EngineSession(shared)
    assert a._lock is b._lock
    inside, release, order, errors = threading.Event(), threading.Event(), [], []

    def second():
        try:
            inside.wait(5)
            with b.bound() as be:
                order.append("b")
                assert be._bound_session is b
        except Exception as exc:       # pragma: no cover
            errors.append(exc)

    t = threading.Thread(target=second)
    t.start()
    with a.bound():
        inside.set()
        t.join(0.3)                                 # the second session is waiting for the lock, not raising
        assert t.is_alive() and not errors
        order.append("a")
    t.join(5)
    assert not errors and order == ["a", "b"]
