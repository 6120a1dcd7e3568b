"""profiles/<tag>_parity.txt from gpurun_out/parity_metrics.json (written by tests/conftest.py::results_log during
`pytest -m gpu`).  Usage: python tools/parity_summary.py r03 gpurun_out/parity_metrics_full.json "<header line>" """
import json
import sys

tag, src = sys.argv[1], sys.argv[2]
header = sys.argv[3] if len(sys.argv) > 3 else ""
d = json.load(open(src))
rows = [
    ("full SDXL UNet, 512^2, B=1 (rel-L2 vs fp32 oracle)", "unet_full_B1_L64_rel_l2"),
    ("full SDXL UNet, 512^2, B=2 worst sample (LayerNorms folded: the anchor program)", "unet_full_B2_L64_worst_rel_l2"),
    ("full SDXL UNet, 512^2, B=17 worst sample (the benchmark's program)", "unet_full_B17_L64_worst_rel_l2"),
    ("full SDXL UNet, 1024^2, B=2 worst sample (cfg 3)", "unet_full_B2_L128_worst_rel_l2"),
    ("full VAE decode, 512^2, B=17 (GroupNorm statistics from the conv epilogues, narrow conv_out, 16-byte stores)", "vae_full_B17_L64"),
    ("full VAE decode, 1024^2", "vae_full_L128"),
    ("cfg-2 transition (17 frames, 512^2, hipGraphs + frontier 16) vs sequential fp32 oracle", "transition_cfg2_full"),
    ("SDXL-base multi-level transition at FULL width (CFG 4.0, encoded-\"\" negative, 6 steps, levels 3/3/4/5) vs oracle", "transition_base_multilevel_full_width"),
    ("tiny-config transition, turbo", "transition_turbo"), ("tiny-config transition, base (CFG)", "transition_base"),
    ("chained transitions (swap_forward + recycle_img1) vs oracle", "chained_transitions"),
    ("branch1 crossfeed (0.3, 0.5, 0.5) on the native pipe (sequential / frontier) vs oracle", "branch1_crossfeed_native"),
    ("latent2image(output_type=\"np\"): unquantised float image", "latent2image_np"),
    ("encode_prompt 4-tuple vs the transformers assembly (negative None / \"\" / string / list + holder default), worst rel-L2", "encode_prompt_4tuple_vs_transformers_assembly"),
    ("fused wavefront at 1024^2 (65,536-element slerps) vs the sequential engine", "wavefront_L128"),
    ("dead-step elision: UNet samples [default, elided], frames bit-identical", "dead_step_elision"),
    ("VAE with the residual stream forced beyond fp16 range, conv_in / fp16x2^-4", "vae_large_act_conv_in_f16s"),
    ("... late / fp16x2^-4", "vae_large_act_late_f16s"), ("... conv_in / fp32 stream", "vae_large_act_conv_in_f32"),
    ("... late / fp32 stream", "vae_large_act_late_f32"),
    ("slerp kernels fed the reference's golden bit patterns", "slerp_gpu_vs_reference_golden"),
    ("HF-layout safetensors -> from_safetensors -> facade", "safetensors_roundtrip"),
    ("CLIP-L text tower vs transformers.CLIPTextModel (CPU fp32, seeded init)", "clip_text_clip_l"),
    ("OpenCLIP-bigG text tower vs transformers.CLIPTextModelWithProjection", "clip_text_openclip_bigg"),
    ("branch farm: two native ranks on one GPU (frontier 8)", "farm_native_2ranks"),
    ("... frontier 1 (rank 1 owns no mid branch)", "farm_native_2ranks_frontier1"), ("... frontier 3 (2 / 1 split)", "farm_native_2ranks_frontier3"),
    ("branch farm on RCCL, world 1", "farm_rccl_world1"),
    ("cfg 3 stated tree (30 steps, depth 0.5, 15 branches, guidance 4.0) at tiny width, frontier 1, vs tests/golden/configs.json", "cfg3_stated_tree_frontier1"),
    ("... frontier 16", "cfg3_stated_tree_frontier16"),
    ("cfg 4 stated tree (Turbo, 64 branches), sequential engine", "cfg4_stated_tree_frontier1"),
    ("... frontier 64 (fused wavefront + virtual gaps)", "cfg4_stated_tree_frontier64"),
    ("cfg 5: 6-prompt chain (swap_forward + recycle_img1) through replay.run_multi_transition, frontier 1", "cfg5_chain_frontier1"),
    ("... frontier 16", "cfg5_chain_frontier16"),
    ("get_state_dict -> yml_save -> load_state_dict -> run_transition on the native pipe", "state_round_trip_native"),
    ("two-stage speculation vs all-at-once, the benchmark's metric", "two_stage_speculation_skew0.0"),
    ("... skewed metric exp(3 x position)", "two_stage_speculation_skew3.0"),
]
out = [header] if header else []
out.append(f"# Source: {src} written by tests/conftest.py::results_log ({len(d)} entries); selection below.\n")
for label, key in rows:
    if key in d:
        out.append(f"{label}: {json.dumps(d[key])}")
kern = {k: v for k, v in d.items() if isinstance(v, dict) and "rel_l2" in v and "bound_abs" in v}
if kern:
    worst = max(kern.items(), key=lambda kv: kv[1]["rel_l2"])
    out.append(f"\nkernel cases with a rel-L2 entry: {len(kern)}; worst GEMM / conv / attention / norm case: {worst[0]} rel-L2 {worst[1]['rel_l2']:.2e}")
open(f"profiles/{tag}_parity.txt", "w").write("\n".join(out) + "\n")
print("\n".join(out))
