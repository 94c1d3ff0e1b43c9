"""CPU tests of the drop-in boundary: the classes importable under the reference's import paths have the
reference's argument lists (committed ast fixture tests/golden/reference_signatures.json, regenerated live when
/root/reference is present), state-dict keys of the plugin modules match the reference's, from_pretrained / save_pretrained
round-trip a diffusers-layout directory, reference-style argument errors are raised, and nothing computes on the CPU."""
import inspect
import json
import os

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLD = os.path.join(ROOT, "tests", "golden", "reference_signatures.json")
TINY = dict(block_out_channels=(64, 128, 256), transformer_layers_per_block=(1, 1, 2), num_attention_heads=(1, 2, 4),
            cross_attention_dim=128, addition_time_embed_dim=32, projection_class_embeddings_input_dim=64 + 6 * 32,
            encoder_hid_dim=128, sample_size=16,
            resampler=dict(dim=128, depth=2, dim_head=64, heads=2, num_queries=16, ff_mult=4))


def _ours():
    from ip_adapter.attention_processor import AttnProcessor2_0, IPAttnProcessor2_0
    from ip_adapter.resampler import Resampler
    from src.tryon_pipeline import StableDiffusionXLInpaintPipeline as P
    from src.unet_hacked_garmnet import UNet2DConditionModel as G
    from src.unet_hacked_tryon import UNet2DConditionModel as T
    return {"ip_adapter/attention_processor.py:AttnProcessor2_0": AttnProcessor2_0,
            "ip_adapter/attention_processor.py:IPAttnProcessor2_0": IPAttnProcessor2_0,
            "ip_adapter/resampler.py:Resampler": Resampler, "src/unet_hacked_tryon.py:UNet2DConditionModel": T,
            "src/unet_hacked_garmnet.py:UNet2DConditionModel": G,
            "src/tryon_pipeline.py:StableDiffusionXLInpaintPipeline": P}


def _argnames(fn):
    out = []
    for n, p in inspect.signature(fn).parameters.items():
        out.append(("*" if p.kind is p.VAR_POSITIONAL else "**" if p.kind is p.VAR_KEYWORD else "") + n)
    return out


def test_call_surface_matches_reference_signatures():
    ref = json.load(open(GOLD))
    ours = _ours()
    # constructors of the two UNets / set_attn_processor differ by design only in private trailing arguments
    for key, spec in ref.items():
        cls_key, meth = key.rsplit(".", 1)
        fn = getattr(ours[cls_key], meth)
        got = _argnames(fn)
        want = spec["args"]
        if meth == "set_attn_processor":
            assert got[:2] == want[:2], (key, got, want)
        else:
            assert got == want, (key, got, want)


@pytest.mark.skipif(not os.path.isdir("/root/reference/src"), reason="reference checkout not present")
def test_signature_fixture_is_current():
    from oracle.make_golden import reference_signatures
    assert reference_signatures() == json.load(open(GOLD))


@pytest.mark.skipif(not os.path.isfile("/root/reference/ip_adapter/resampler.py"), reason="reference checkout not present")
def test_resampler_state_dict_keys_match_reference_module():
    import importlib.util
    spec = importlib.util.spec_from_file_location("ref_resampler_b", "/root/reference/ip_adapter/resampler.py")
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    kw = dict(dim=128, depth=2, dim_head=64, heads=2, num_queries=16, embedding_dim=96, output_dim=160, ff_mult=4)
    from ip_adapter.resampler import Resampler
    ref, ours = m.Resampler(**kw), Resampler(**kw)
    assert {k: tuple(v.shape) for k, v in ref.state_dict().items()} == {k: tuple(v.shape) for k, v in ours.state_dict().items()}
    ours.load_state_dict(ref.state_dict(), strict=True)


def _tiny_models(dtype=torch.float16):
    from idm_vton_amd import config as pc
    from idm_vton_amd.boundary.unet import GarmentUNet2DConditionModel, TryonUNet2DConditionModel
    from idm_vton_amd.boundary.vae import AutoencoderKL
    tcfg = pc.UNetConfig(mode="tryon", in_channels=13, **TINY)
    gcfg = pc.UNetConfig(mode="garmnet", in_channels=4, addition_embed_type=None, encoder_hid_dim_type=None, **TINY)
    vcfg = pc.VAEConfig(block_out_channels=(32, 32, 32, 32), layers_per_block=1)
    t = TryonUNet2DConditionModel(tcfg, torch_dtype=dtype)
    g = GarmentUNet2DConditionModel(gcfg, torch_dtype=dtype)
    v = AutoencoderKL(vcfg, torch_dtype=dtype)
    for m, seed in ((t, 1), (g, 2), (v, 3)):
        gen = torch.Generator().manual_seed(seed)
        with torch.no_grad():
            for p in m.parameters():
                p.copy_((torch.randn(p.shape, generator=gen) * 0.02).to(p.dtype))
    return t, g, v


def test_unet_attribute_surface_and_state_dict_keys():
    from idm_vton_amd import config as pc
    t, g, v = _tiny_models()
    assert t.config.in_channels == 13 and g.config.in_channels == 4 and t.config.time_cond_proj_dim is None
    assert t.config.encoder_hid_dim_type == "ip_image_proj" and callable(t.encoder_hid_proj)
    assert t.add_embedding.linear_1.in_features == TINY["projection_class_embeddings_input_dim"]     # tryon_pipeline.py:1049
    assert t.dtype == torch.float16 and str(t.device) == "cpu"
    assert set(t.state_dict()) == {n for n, _ in pc.unet_param_shapes(t.cfg)}
    assert set(g.state_dict()) == {n for n, _ in pc.unet_param_shapes(g.cfg)}
    assert any(k.endswith("attn2.processor.to_k_ip.weight") for k in t.state_dict())
    procs = t.attn_processors
    assert len(procs) == 2 * 17 and all(k.endswith(".processor") for k in procs)                      # 17 blocks x (attn1, attn2)
    with pytest.raises(ValueError, match="number of processors"):
        t.set_attn_processor({k: p for k, p in list(procs.items())[:3]})
    t.set_attn_processor(procs)


def _round_trip(tmp_path):
    from idm_vton_amd.boundary.scheduler import DDPMScheduler
    from src.tryon_pipeline import StableDiffusionXLInpaintPipeline
    from src.unet_hacked_garmnet import UNet2DConditionModel as G
    from src.unet_hacked_tryon import UNet2DConditionModel as T
    t, g, v = _tiny_models()
    root = str(tmp_path)
    t.save_pretrained(os.path.join(root, "unet"))
    g.save_pretrained(os.path.join(root, "unet_encoder"))
    v.save_pretrained(os.path.join(root, "vae"))
    DDPMScheduler().save_pretrained(os.path.join(root, "scheduler"))
    t2 = T.from_pretrained(root, subfolder="unet", torch_dtype=torch.float16)                         # inference.py:239-243
    g2 = G.from_pretrained(root, subfolder="unet_encoder", torch_dtype=torch.float16)                 # inference.py:250-254
    for a, b in ((t, t2), (g, g2)):
        sa, sb = a.state_dict(), b.state_dict()
        assert set(sa) == set(sb) and all(torch.equal(sa[k], sb[k]) for k in sa)
    with pytest.raises(EnvironmentError):
        T.from_pretrained(root, subfolder="nope")
    pipe = StableDiffusionXLInpaintPipeline.from_pretrained(root, unet=t2, unet_encoder=g2, text_encoder=None, text_encoder_2=None,
                                                            tokenizer=None, tokenizer_2=None, image_encoder=None,
                                                            feature_extractor=None, torch_dtype=torch.float16)   # inference.py:316-329
    assert pipe.vae_scale_factor == 8 and type(pipe.scheduler).__name__ == "DDPMScheduler" and str(pipe.device) == "cpu"
    # app.py:111-125 builds the pipeline without unet_encoder and assigns it afterwards
    pipe2 = StableDiffusionXLInpaintPipeline.from_pretrained(root, unet=t2, text_encoder=None, text_encoder_2=None, tokenizer=None,
                                                             tokenizer_2=None, image_encoder=None, feature_extractor=None)
    assert pipe2.unet_encoder is not None                                                             # found on disk here
    return pipe


def test_from_pretrained_round_trip(tmp_path):
    _round_trip(tmp_path)


def test_memory_toggles_of_the_call_surface_exist_and_keep_the_model_intact(tmp_path):
    """enable/disable_vae_slicing|tiling (/root/reference/src/tryon_pipeline.py:427-457), set_attention_slice, fuse/unfuse_qkv_projections,
    set_default_attn_processor (src/unet_hacked_tryon.py:854-1004): a caller that toggles them must not hit AttributeError; they are
    honest no-ops (already blockwise / already fused) that leave weights and processors usable."""
    pipe = _round_trip(tmp_path)
    assert pipe.vae.use_slicing is False and pipe.vae.use_tiling is False    # diffusers' defaults, readable before any toggle
    pipe.enable_vae_slicing(); assert pipe.vae.use_slicing is True
    pipe.disable_vae_slicing(); assert pipe.vae.use_slicing is False
    pipe.enable_vae_tiling(); assert pipe.vae.use_tiling is True
    pipe.disable_vae_tiling(); assert pipe.vae.use_tiling is False
    t = pipe.unet
    before = {k: v.clone() for k, v in t.state_dict().items()}
    procs = dict(t.attn_processors)
    n_attn = len(procs)
    for s in ("auto", "max", 1, [1] * n_attn):
        t.set_attention_slice(s)
    with pytest.raises(ValueError, match="slice_size"):
        t.set_attention_slice("half")
    with pytest.raises(ValueError, match="different attention layers"):      # reference :921-925: one entry per sliceable layer
        t.set_attention_slice([2, 2])
    with pytest.raises(ValueError, match="has to be smaller or equal"):      # reference :927-931
        t.set_attention_slice(10 ** 6)
    with pytest.raises(ValueError, match="Cannot call `set_default_attn_processor`"):   # reference :854-865: IP processors are in neither set
        t.set_default_attn_processor()
    assert all(t.attn_processors[k] is procs[k] for k in procs)
    t.fuse_qkv_projections()
    assert t.original_attn_processors.keys() == procs.keys()
    t.unfuse_qkv_projections()
    assert all(t.attn_processors[k] is procs[k] for k in procs)
    after = t.state_dict()
    assert set(before) == set(after) and all(torch.equal(before[k], after[k]) for k in before)
    with pytest.raises(NotImplementedError, match="FreeU"):
        t.enable_freeu(0.9, 0.2, 1.2, 1.4)
    t.disable_freeu()
    g = pipe.unet_encoder
    g.set_default_attn_processor()
    assert all(type(p).__name__ == "AttnProcessor2_0" for p in g.attn_processors.values())


def test_pipeline_argument_errors_and_no_cpu_path(tmp_path):
    pipe = _round_trip(tmp_path)
    B, H, W = 1, 128, 128
    z = lambda *s: torch.zeros(*s)
    kw = dict(prompt_embeds=z(B, 77, 128), negative_prompt_embeds=z(B, 77, 128), pooled_prompt_embeds=z(B, 64),
              negative_pooled_prompt_embeds=z(B, 64), num_inference_steps=2, strength=1.0, pose_img=z(B, 3, H, W),
              text_embeds_cloth=z(B, 77, 128), cloth=z(B, 3, H, W), mask_image=z(B, 1, H, W), image=z(B, 3, H, W), height=H,
              width=W, guidance_scale=2.0, ip_adapter_image=z(B, 3, 224, 224))
    with pytest.raises(ValueError, match="divisible by 8"):
        pipe(**{**kw, "height": 100})
    with pytest.raises(ValueError, match="Cannot forward both `prompt`"):
        pipe(prompt="a shirt", **kw)
    with pytest.raises(ValueError, match="strength"):
        pipe(**{**kw, "strength": 1.5})
    with pytest.raises(ValueError, match="same shape"):
        pipe(**{**kw, "negative_prompt_embeds": z(B, 70, 128)})
    with pytest.raises(ValueError, match="cannot be undefined"):
        pipe(**{**kw, "image": None})
    with pytest.raises(ValueError, match="number of pipeline"):      # reference :1568-1572: int(2 * 0.3) = 0 steps left
        pipe(**{**kw, "strength": 0.3})
    # the product has no CPU path: modules on the CPU refuse to run instead of falling back
    with pytest.raises(RuntimeError, match="GPU only"):
        pipe(**kw)


def test_scheduler_matches_oracle_coefficients():
    """boundary DDPMScheduler.step == oracle scheduler step on CPU tensors."""
    from idm_vton_amd.boundary.scheduler import DDIMScheduler, DDPMScheduler
    from oracle.scheduler import Scheduler
    g = torch.Generator().manual_seed(0)
    x, eps, nz = (torch.randn(2, 4, 8, 8, generator=g) for _ in range(3))
    for cls, kind in ((DDPMScheduler, "ddpm"), (DDIMScheduler, "ddim")):
        s, o = cls(), Scheduler(kind)
        s.set_timesteps(30)
        o.set_timesteps(30)
        assert s.timesteps.tolist() == list(o.timesteps)
        t = int(s.timesteps[3])
        c_x, c_eps, sigma = s._impl.coeffs(t)
        ours = c_x * x + c_eps * eps + sigma * nz
        ref = o.step(eps, t, x, noise=nz)
        assert torch.allclose(ours, ref, atol=1e-5), (kind, (ours - ref).abs().max())
