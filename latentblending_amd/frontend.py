"""The logic of the reference's Gradio front-end without the widgets.

``/root/reference/latentblending/gradio_ui.py`` has two classes under its ``gr.Blocks``: ``BlendingVariableHolder`` (:94-262 -
one user's preview images, the list of key frames of the movie being edited, the JSON it saves after every edit, and
``generate_movie``: the swap_forward / recycle_img1 chain over that list) and ``MultiUserRouter`` (:29-90 - engines per model
name, one holder per user id, every UI callback forwarded by user id).  Both exist here with the same method names, the
same state fields and the same files written, so that a UI layer (or a test, or a REST handler) drives them exactly as
the reference's button bindings do (:332-348).  Differences, each deliberate:

* users are isolated: a holder works through an ``EngineSession`` (session.py) - prompts, seeds, size, tree and stored
  trajectories of one user never stay on the shared engine between two calls (the reference's holders all mutate the one
  engine of their model: ``register_new_user`` of a second user changes the first user's output size, gradio_ui.py:48-53);
* the two ``*_selected`` callbacks take an index (or any object with an ``index`` attribute, which ``gr.SelectData`` is);
* ``img_movie_later`` on the LAST image and ``img_movie_delete`` with an index one past the end are refused with the
  reference's own message instead of raising ``IndexError`` (its guards are off by one, :197, :206);
* the engines are handed in (``MultiUserRouter(engines={...})``) or built through the ``diffusers`` facade with the
  reference's two model names; ``launch_ui`` builds the reference's Blocks layout when ``gradio`` is importable (it is
  not in this image) and says so otherwise.

Host-side control only; the hot path below it is ``BlendingEngine.run_transition`` / ``compute_latents1``.
"""
from __future__ import annotations

import datetime
import json
import os
import tempfile
import uuid
from typing import Dict, List, Optional

import numpy as np

from .session import EngineSession, SessionRouter

LIST_MODELS = ["stabilityai/sdxl-turbo", "stabilityai/stable-diffusion-xl-base-1.0"]        # gradio_ui.py:36


def _index_of(data) -> int:
    return int(getattr(data, "index", data))


class BlendingVariableHolder:
    """One user's UI state (gradio_ui.py:94-131) over an ``EngineSession``."""

    def __init__(self, be, session: Optional[EngineSession] = None, dp_out: Optional[str] = None):
        self.be = be
        self.session = session if session is not None else EngineSession(be)
        self.dp_out = dp_out or "."                 # where movie_*.mp4 / movie_*.json / tmp_part_*.mp4 go (the reference: the cwd)
        # UI defaults
        self.seed1 = 420
        self.seed2 = 420
        self.prompt1 = ""
        self.prompt2 = ""
        self.nmb_preview_images = 4
        # vars
        self.prompt = None
        self.negative_prompt = None
        self.list_seeds: List[int] = []
        self.idx_movie = 0
        self.list_images_preview: List[str] = []
        self.data: List[Dict] = []
        self.idx_img_preview_selected = None
        self.idx_img_movie_selected = None
        self.jpg_quality = 80
        self.fp_movie = ''
        self.fp_json = ''

    # ---- selections (gradio_ui.py:133-139) ----
    def preview_img_selected(self, data, button=None):
        self.idx_img_preview_selected = _index_of(data)

    def movie_img_selected(self, data, button=None):
        self.idx_img_movie_selected = _index_of(data)

    # ---- previews (gradio_ui.py:141-161): nmb_preview_images renders of ONE prompt under random seeds ----
    def compute_imgs(self, prompt, negative_prompt):
        from PIL import Image
        self.prompt = prompt
        self.negative_prompt = negative_prompt
        self.list_seeds = []
        self.list_images_preview = []
        self.idx_img_preview_selected = None
        with self.session.bound() as be:
            be.set_prompt1(prompt)
            be.set_prompt2(prompt)
            be.set_negative_prompt(negative_prompt)
            for _ in range(self.nmb_preview_images):
                seed = int(np.random.randint(0, np.iinfo(np.int32).max))
                be.seed1 = seed
                self.list_seeds.append(seed)
                img = be.compute_latents1(return_image=True)
                if not isinstance(img, Image.Image):
                    img = Image.fromarray(np.asarray(img))
                fp = os.path.join(tempfile.gettempdir(), f"image_{uuid.uuid4()}.jpg")
                img.convert("RGB").save(fp, quality=self.jpg_quality, optimize=True)
                self.list_images_preview.append(fp)
        return self.list_images_preview

    def get_list_images_movie(self):
        return [entry["preview_image"] for entry in self.data]

    # ---- the movie being edited (gradio_ui.py:168-218) ----
    def init_new_movie(self):
        stamp = datetime.datetime.now().strftime("%y%m%d_%H%M")
        self.fp_movie = os.path.join(self.dp_out, "movie_" + stamp + ".mp4")
        self.fp_json = os.path.join(self.dp_out, "movie_" + stamp + ".json")

    def write_json(self):
        with self.session.bound() as be:
            header = {"settings": "sdxl", "width": be.dh.width_img, "height": be.dh.height_img,
                      "num_inference_steps": be.dh.num_inference_steps}
        with open(self.fp_json, 'w') as f:
            json.dump([header] + list(self.data), f, indent=4)

    def add_image_to_video(self):
        if self.prompt is None:
            print("Cannot take because no prompt was set!")
            return self.get_list_images_movie()
        if self.idx_img_preview_selected is None or not 0 <= self.idx_img_preview_selected < len(self.list_seeds):
            print("Cannot take because no preview image is selected!")          # (the reference indexes its lists with None here)
            return self.get_list_images_movie()
        if self.idx_movie == 0:
            self.init_new_movie()
        self.data.append({"iteration": self.idx_movie,
                          "seed": self.list_seeds[self.idx_img_preview_selected],
                          "prompt": self.prompt,
                          "negative_prompt": self.negative_prompt,
                          "preview_image": self.list_images_preview[self.idx_img_preview_selected]})
        self.write_json()
        self.idx_movie += 1
        return self.get_list_images_movie()

    def img_movie_delete(self):
        if self.idx_img_movie_selected is not None and 0 <= self.idx_img_movie_selected < len(self.data):
            del self.data[self.idx_img_movie_selected]
            self.idx_img_movie_selected = None
        else:
            print(f"Invalid movie image index for deletion: {self.idx_img_movie_selected}")
        return self.get_list_images_movie()

    def img_movie_later(self):
        i = self.idx_img_movie_selected
        if i is not None and 0 <= i < len(self.data) - 1:
            self.data[i], self.data[i + 1] = self.data[i + 1], self.data[i]
            self.idx_img_movie_selected = None
        else:
            print("Cannot move the image later in the sequence.")
        return self.get_list_images_movie()

    def img_movie_earlier(self):
        i = self.idx_img_movie_selected
        if i is not None and 0 < i < len(self.data):
            self.data[i - 1], self.data[i] = self.data[i], self.data[i - 1]
            self.idx_img_movie_selected = None
        else:
            print("Cannot move the image earlier in the sequence.")
        return self.get_list_images_movie()

    # ---- gradio_ui.py:222-262: the chain over the edited list, one part per segment, concatenated ----
    def generate_movie(self, t_per_segment=10, fps: int = 30):
        from .replay import run_multi_transition
        if len(self.data) < 2:
            raise ValueError("generate_movie: the movie needs at least two images (add_image_to_video)")
        if not self.fp_movie:
            self.init_new_movie()
        with self.session.bound() as be:
            run_multi_transition(be, [it["prompt"] for it in self.data], [it["seed"] for it in self.data], self.fp_movie,
                                 duration_single_trans=t_per_segment,
                                 list_negative_prompts=[it["negative_prompt"] for it in self.data], fps=fps, dp_parts=self.dp_out)
        print(f"DONE! MOVIE SAVED IN {self.fp_movie}")
        return self.fp_movie


class MultiUserRouter:
    """gradio_ui.py:29-90: engines per model, a holder per user id, every callback forwarded by user id."""

    def __init__(self, do_compile: bool = False, engines: Optional[Dict[str, object]] = None, list_models=None,
                 dp_out: Optional[str] = None):
        self.user_blendingvariableholder: Dict[str, BlendingVariableHolder] = {}
        self.do_compile = do_compile
        self.dp_out = dp_out
        if engines is not None:
            self.dict_blendingengines = dict(engines)
            self.list_models = list(self.dict_blendingengines)
        else:
            self.list_models = list(list_models or LIST_MODELS)
            self.init_models()
        self._sessions = SessionRouter(self.dict_blendingengines)

    def init_models(self):
        """The reference's loop (gradio_ui.py:40-47) through the ``diffusers`` facade of this repo (native pipes)."""
        import torch
        from diffusers import AutoPipelineForText2Image
        from .blending_engine import BlendingEngine
        self.dict_blendingengines = {}
        for m in self.list_models:
            pipe = AutoPipelineForText2Image.from_pretrained(m, torch_dtype=torch.float16, variant="fp16")
            pipe.to("cuda")
            self.dict_blendingengines[m] = BlendingEngine(pipe, do_compile=self.do_compile)

    def register_new_user(self, model, width, height):
        user_id = self._sessions.register_new_user(model, int(width), int(height))
        self.user_blendingvariableholder[user_id] = BlendingVariableHolder(self.dict_blendingengines[model],
                                                                           self._sessions.session(user_id), dp_out=self.dp_out)
        return user_id

    def user_overflow_protection(self):
        pass


def _forward(name):
    """Router method ``name(user_id, *args)`` -> the user's holder method ``name(*args)`` (gradio_ui.py:58-90: thirteen one-line forwards)."""
    def call(self, user_id, *args):
        return getattr(self.user_blendingvariableholder[user_id], name)(*args)
    call.__name__ = name
    call.__doc__ = f"``BlendingVariableHolder.{name}`` of user ``user_id``."
    return call


for _name in ("preview_img_selected", "movie_img_selected", "compute_imgs", "get_list_images_movie", "init_new_movie", "write_json",
              "add_image_to_video", "img_movie_delete", "img_movie_later", "img_movie_earlier", "generate_movie"):
    setattr(MultiUserRouter, _name, _forward(_name))
del _name


def launch_ui(mur: MultiUserRouter, nmb_preview_images: int = 4, server_name: Optional[str] = None, launch: bool = True):
    """A ``gradio`` page over ``mur`` with the reference page's controls (gradio_ui.py:286-350): a set-up row (model, size, "start
    session" -> user id), the prompt row with the preview gallery, the movie gallery with its delete / earlier / later buttons, and
    "generate movie".  Every control is bound to the router method of the same purpose.  Needs the ``gradio`` package."""
    try:
        import gradio as gr
    except ImportError as exc:          # (not installable in the build image: the logic above is what is tested)
        raise ImportError("latentblending_amd.frontend.launch_ui needs `gradio`; MultiUserRouter / BlendingVariableHolder "
                          "carry the UI's logic without it") from exc

    def gallery(columns):
        return gr.Gallery(show_label=False, columns=[columns], rows=[1], object_fit="contain", height="auto", allow_preview=False,
                          interactive=False)

    with gr.Blocks() as demo:
        with gr.Accordion("Setup", open=True), gr.Row():
            model = gr.Dropdown(mur.list_models, value=mur.list_models[0], label="model")
            size = [gr.Slider(256, 2048, 512, step=128, label=name, interactive=True) for name in ("width", "height")]
            user_id = gr.Textbox(label="user id (filled automatically)", interactive=False)
            start = gr.Button("start session", variant="primary")
        with gr.Accordion("Latent Blending (open after 'start session')", open=False):
            with gr.Row():
                prompt, negative = gr.Textbox(label="prompt"), gr.Textbox(label="negative prompt")
                compute = gr.Button("generate preview images", variant="primary")
                take = gr.Button("add selected image to video", variant="primary")
            with gr.Row():
                previews = gallery(nmb_preview_images)
            gr.Markdown("Images of the movie, in order:")
            with gr.Row():
                film = gallery(20)
            with gr.Row():
                edits = {name: gr.Button(label) for name, label in (("img_movie_delete", "delete selected image"),
                                                                     ("img_movie_earlier", "move image to earlier time"),
                                                                     ("img_movie_later", "move image to later time"))}
            with gr.Row():
                render = gr.Button("generate movie", variant="primary")
                seconds = gr.Slider(1, 30, 10, step=0.1, label="time per segment", interactive=True)
            video = gr.Video()
        start.click(mur.register_new_user, inputs=[model] + size, outputs=user_id)
        compute.click(mur.compute_imgs, inputs=[user_id, prompt, negative], outputs=previews)
        take.click(mur.add_image_to_video, user_id, film)
        previews.select(mur.preview_img_selected, user_id, None)
        film.select(mur.movie_img_selected, user_id, None)
        for name, button in edits.items():
            button.click(getattr(mur, name), user_id, film)
        render.click(mur.generate_movie, [user_id, seconds], video)
    if launch:
        kw = {} if server_name is None else {"server_name": server_name}
        demo.launch(share=False, inbrowser=True, inline=False, **kw)
    return demo
