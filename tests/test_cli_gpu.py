"""End-to-end plumbing on the GPU: synthetic on-disk datasets (two action domains) -> train_multi for a few optimizer
steps (sampler + device collator + fused trainer) -> checkpoint -> generate.py rollout -> output files in the
reference's format."""
import json

import numpy as np
import pytest
import torch

from hma_amd import generate as hgen
from hma_amd import train_multi as htrain
from hma_amd.data import RawTokenDataset, write_token_dataset

pytestmark = pytest.mark.gpu


def _dataset(path, name, n, action_dim, seed):
    rng = np.random.default_rng(seed)
    tokens = rng.integers(0, 262144, size=(n, 16, 16), dtype=np.uint32)
    seg = np.repeat(np.arange(n // 20 + 1), 20)[:n].astype(np.int32)
    write_token_dataset(path / name, tokens, seg, rng.standard_normal((n, action_dim)).astype(np.float32), name=name)
    return tokens


def test_train_then_generate(tmp_path):
    tok_a = _dataset(tmp_path, "domA", 80, 7, 0)
    _dataset(tmp_path, "domB", 60, 5, 1)
    cfg = {"num_layers": 2, "num_heads": 8, "use_actions": True, "d_model": 256, "T": 4, "S": 256, "image_vocab_size": 262144,
           "use_mup": False, "action_network": "concat+modulate", "num_factored_vocabs": 2, "qkv_bias": False, "proj_bias": True,
           "mlp_bias": True, "qk_norm": False, "attn_drop": 0.0, "mlp_ratio": 4.0, "mlp_drop": 0.0}
    (tmp_path / "cfg.json").write_text(json.dumps(cfg))
    steps = htrain.main(["--train_data_dir", str(tmp_path / "domA"), str(tmp_path / "domB"), "--genie_config", str(tmp_path / "cfg.json"),
                         "--window_size", "4", "--stride", "1", "--per_device_train_batch_size", "2", "--max_train_steps", "4",
                         "--num_warmup_steps", "2", "--output_dir", str(tmp_path / "out"), "--seed", "0", "--log_every", "2"])
    assert steps == 4
    ckpt = tmp_path / "out" / "step_4"
    assert (ckpt / "config.json").exists() and (ckpt / "model.safetensors").exists()
    saved = json.load(open(ckpt / "config.json"))
    assert saved["action_domains"] == ["domA", "domB"] and saved["d_actions"] == [7, 5]

    out = hgen.main(["--val_data_dir", str(tmp_path / "domA"), "--checkpoint_dir", str(ckpt), "--output_dir", str(tmp_path / "gen"),
                     "--num_prompt_frames", "2", "--window_size", "4", "--maskgit_steps", "2", "--batch_size", "2", "--max_example", "2",
                     "--add_action_input"])
    # [prompt (2) | generated (2) | ground truth (2)] frames per example
    assert out.shape[1:] == (6, 16, 16)
    meta = json.load(open(tmp_path / "gen" / "metadata.json"))
    # trained_steps is scheduler.bin's `_step_count` like the reference reports it (generate.py:80-84): optimizer steps + 1
    assert meta["num_images"] == 6 and meta["t"] == 4 and meta["trained_steps"] == 5 and meta["token_dtype"] == "uint32"
    raw = np.fromfile(tmp_path / "gen" / "video.bin", dtype=np.uint32).reshape(-1, 6, 16, 16)
    assert raw.shape[0] == out.shape[0]
    first = RawTokenDataset(tmp_path / "domA", window_size=4, compute_stride_from_freq_table=False)[0]["input_ids"].reshape(4, 16, 16)
    assert np.array_equal(raw[0, :2], first[:2].numpy().astype(np.uint32))        # prompt frames kept
    assert np.array_equal(raw[0, 4:], first[2:].numpy().astype(np.uint32))        # ground truth appended
    assert (raw[:, 2:4] < 262144).all()                                           # generated ids are real tokens
    assert np.array_equal(tok_a[:2], raw[0, :2])


def test_train_then_generate_continuous(tmp_path):
    """The MAR data path end to end (`--model_type continuous`, `--use_feature`): VAE-latent datasets in the reference's layout
    -> RawFeatureDataset + get_maskgit_collator_feature -> MarTrainer steps -> checkpoint -> STMAR rollout -> float32
    `video.bin` laid out (b, t, c, h, w) (hma/train_multi.py:756-775, hma/generate.py:108-117, 187-189)."""
    from hma_amd.data import write_feature_dataset
    from tests.golden.stmar_cfg import CFG

    rng = np.random.default_rng(3)
    lat = {}
    for name, n, adim in (("domA", 40, 7), ("domB", 30, 5)):
        lat[name] = rng.standard_normal((n, 4, 32, 32)).astype(np.float16)
        seg = np.repeat(np.arange(n // 10 + 1), 10)[:n].astype(np.int32)
        write_feature_dataset(tmp_path / name, lat[name], seg, rng.standard_normal((n, adim)).astype(np.float32), name=name)
    cfg = dict(CFG, use_actions=True, num_sampling_steps="5", maskgit_steps=2)
    (tmp_path / "cfg.json").write_text(json.dumps(cfg))
    steps = htrain.main(["--train_data_dir", str(tmp_path / "domA"), str(tmp_path / "domB"), "--genie_config", str(tmp_path / "cfg.json"),
                         "--model_type", "continuous", "--window_size", "3", "--stride", "1", "--per_device_train_batch_size", "2",
                         "--max_train_steps", "3", "--num_warmup_steps", "1", "--output_dir", str(tmp_path / "out"), "--seed", "0",
                         "--log_every", "1"])
    assert steps == 3
    ckpt = tmp_path / "out" / "step_3"
    assert (ckpt / "config.json").exists() and (ckpt / "model.safetensors").exists()
    out = hgen.main(["--val_data_dir", str(tmp_path / "domA"), "--checkpoint_dir", str(ckpt), "--output_dir", str(tmp_path / "gen"),
                     "--num_prompt_frames", "1", "--window_size", "3", "--batch_size", "2", "--max_example", "2", "--add_action_input",
                     "--use_feature", "--temperature", "1.0"])
    assert out.shape[1:] == (5, 4, 32, 32) and torch.isfinite(out).all()          # [prompt 1 | generated 2 | ground truth 2], (c, h, w)
    meta = json.load(open(tmp_path / "gen" / "metadata.json"))
    assert meta["num_images"] == 5 and meta["t"] == 3 and meta["token_dtype"] == "float32" and meta["latent_channels"] == 4
    raw = np.fromfile(tmp_path / "gen" / "video.bin", dtype=np.float32).reshape(-1, 5, 4, 32, 32)
    want = lat["domA"][:3].astype(np.float32) * 0.18215                            # SVD_SCALE on load (hma/data.py:416)
    assert np.allclose(raw[0, 0], want[0], atol=1e-6) and np.allclose(raw[0, 3:], want[1:], atol=1e-6)
