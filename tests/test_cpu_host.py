"""CPU-only tests (run in the build container, no GPU): C-ABI surface, host logic, synthetic data invariants."""
import os
import re
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def test_library_loads_and_exports_every_declared_symbol():
    from spokennlp_amd import lib
    hdr = open(os.path.join(ROOT, "include", "amdseg.h")).read()
    declared = set(re.findall(r"\b(amdseg_[a-z0-9_]+)\s*\(", hdr))
    declared -= {"amdseg_stream_t"}
    assert declared, "no declarations parsed"
    l = lib.load()
    for name in sorted(declared):
        assert hasattr(l, name), f"{name} declared in include/amdseg.h but not exported"
    assert set(lib.EXPORTS) == declared, (set(lib.EXPORTS) ^ declared)
    assert l.amdseg_abi_version() == lib.ABI_VERSION
    assert b"shape" in l.amdseg_error_string(1001)


def test_missing_library_fails_loudly(tmp_path):
    from spokennlp_amd import lib
    with pytest.raises(lib.AmdsegError):
        lib.load(str(tmp_path / "nope.so"))


def test_model_refuses_cpu_tensors():
    from transformers import BertConfig
    from spokennlp_amd import lib
    from spokennlp_amd.bert_for_ts import BertWithDAForSentenceLabelingTopicSegmentation as M
    cfg = BertConfig(vocab_size=50, hidden_size=128, num_hidden_layers=1, num_attention_heads=2, intermediate_size=128, num_labels=2)
    m = M(cfg)
    ids = torch.zeros(1, 2, 64, dtype=torch.long)
    with pytest.raises(lib.AmdsegError):
        m(input_ids=ids, labels=torch.full((1, 2, 64), -100))


def test_hf_parameter_names_match_reference_checkpoint_layout():
    from transformers import BertConfig
    from spokennlp_amd.bert_for_ts import BertWithDAForSentenceLabelingTopicSegmentation as M
    cfg = BertConfig(vocab_size=50, hidden_size=128, num_hidden_layers=2, num_attention_heads=2, intermediate_size=256, num_labels=2)
    names = set(dict(M(cfg).named_parameters()))
    for n in ["bert.embeddings.word_embeddings.weight", "bert.encoder.layer.1.attention.self.query.weight",
              "bert.encoder.layer.0.attention.output.LayerNorm.bias", "bert.encoder.layer.1.intermediate.dense.weight",
              "bert.pooler.dense.weight", "loss_calculator.classifier.weight", "loss_calculator.classifier.bias",
              "loss_calculator.tssp.classifier.weight", "loss_calculator.tssp.classifier.bias"]:
        assert n in names, n


def test_flat_params_alias_and_qkv_adjacent():
    from transformers import BertConfig
    from spokennlp_amd.bert_for_ts import BertWithDAForSentenceLabelingTopicSegmentation as M
    from spokennlp_amd.engine import FlatParams
    cfg = BertConfig(vocab_size=50, hidden_size=128, num_hidden_layers=2, num_attention_heads=2, intermediate_size=256, num_labels=2)
    m = M(cfg)
    before = {n: p.detach().clone() for n, p in m.named_parameters()}
    fp = FlatParams(m, torch.device("cpu"))
    for n, p in m.named_parameters():
        assert torch.equal(p.detach(), before[n])
        assert p.data_ptr() == fp.flat_p.data_ptr() + 4 * fp.offsets[n]
        assert p.grad is not None and p.grad.data_ptr() == fp.flat_g.data_ptr() + 4 * fp.offsets[n]
    H = 128
    for i in range(2):
        q = fp.offsets[f"bert.encoder.layer.{i}.attention.self.query.weight"]
        assert fp.offsets[f"bert.encoder.layer.{i}.attention.self.key.weight"] == q + H * H
        assert fp.offsets[f"bert.encoder.layer.{i}.attention.self.value.weight"] == q + 2 * H * H
        qb = fp.offsets[f"bert.encoder.layer.{i}.attention.self.query.bias"]
        assert fp.offsets[f"bert.encoder.layer.{i}.attention.self.value.bias"] == qb + 2 * H
    assert fp.intact()
    # writes through a parameter land in the flat buffer
    with torch.no_grad():
        m.loss_calculator.classifier.bias.fill_(3.0)
    o = fp.offsets["loss_calculator.classifier.bias"]
    assert fp.flat_p[o].item() == 3.0


@pytest.mark.parametrize("L,vocab", [(64, 200), (512, 30523)])
def test_synthetic_batches_satisfy_reference_invariants(L, vocab):
    """the invariants without which the reference heads crash or silently skip."""
    from spokennlp_amd import data
    kw = dict(mean_sents=14, sd_sents=5, mean_boundaries=3, mu_tok=1.6, sigma_tok=0.4) if L == 64 else {}
    docs = data.synth_docs(12, seed=3, vocab=vocab, **kw)
    bs = data.batches_from_docs(docs, L, 4, seed=1, as_torch=False)
    assert len(bs) >= 2
    bos = vocab - 1
    for b in bs:
        for c in data.COLUMNS:
            assert b[c].shape == (4, 2, L) and b[c].dtype == np.int64
        for i in range(4):
            for s in range(2):
                ids, lab = b["input_ids"][i, s], b["labels"][i, s]
                am, seg = b["attention_mask"][i, s], b["extract_eop_segment_ids"][i, s]
                assert ids[0] == data.CLS_ID and lab[0] == -100
                labelled = np.nonzero(lab != -100)[0]
                assert (ids[labelled] == bos).all()
                assert set(lab[labelled].tolist()) <= {0, 1}
                k = len(labelled)
                assert seg[labelled].tolist() == list(range(1, k + 1)) and (np.delete(seg, labelled) == 0).all()
                eidx = b["eop_index_for_aggregate_batch_eop_features"][i, s]
                assert eidx[:k + 1].tolist() == list(range(k + 1)) and (eidx[k + 1:] == 0).all()
                n = int(am.sum())
                assert (am[:n] == 1).all() and (am[n:] == 0).all() and (ids[n:] == data.PAD_ID).all() and (lab[n:] == -100).all()
                is_bos = (ids == bos) & (am == 1)
                stm = b["sent_token_mask"][i, s]
                assert ((stm != -100) == is_bos).all()
                assert (stm[labelled] == (lab[labelled] != 0)).all()
                sll = b["sent_level_labels"][i, s]
                assert sll[0] == -100 and sll[1:1 + int(is_bos.sum())].tolist() == lab[is_bos].tolist()
            if L == 64 or True:
                # TSSP labels sit on every BOS of the augmented half; anchors carry a copy of them
                spo = b["sent_pair_orders"][i, 1]
                da_bos = (b["input_ids"][i, 1] == bos) & (b["attention_mask"][i, 1] == 1)
                assert ((spo != -100) == da_bos).all() and set(spo[da_bos].tolist()) <= {0, 1, 2}
                assert (b["sent_pair_orders"][i, 0] == spo).all()
            assert (b["labels"][i, 0] != -100).sum() >= 1
            # the augmented half is a permutation of the anchor's tokens
            assert sorted(b["input_ids"][i, 0].tolist()) == sorted(b["input_ids"][i, 1].tolist())


def test_dense_batch_shape():
    from spokennlp_amd import data
    b = data.dense_batch(3, L=512, as_torch=False)
    assert b["input_ids"].shape == (3, 2, 512) and (b["attention_mask"] == 1).all()
    assert ((b["labels"][:, 0] != -100).sum(-1) == 20).all()       # 21 sentences, the last one unlabelled


def test_encoder_shape_alignment_rule():
    """EncoderFn pads [B, L] to 64-token blocks and 128-row tiles (engine.py); the rule itself is host logic"""
    from spokennlp_amd.engine import EncoderFn
    for B, L in [(1, 40), (3, 50), (3, 100), (1, 64), (2, 64), (5, 192), (7, 4096), (1, 1)]:
        Bp, Lp = EncoderFn.aligned_shape(B, L)
        assert Lp % 64 == 0 and Lp >= L and Lp - L < 64
        assert Bp >= B and (Bp * Lp) % 128 == 0 and Bp - B <= 1
    assert EncoderFn.aligned_shape(32, 512) == (32, 512)


def test_ctypes_structs_match_the_header_layout(tmp_path):
    """the five structs of the composite entry points are declared twice (include/amdseg.h for C callers, spokennlp_amd/lib.py for ctypes);
    a field added on one side only would shift every pointer behind it.  gcc compiles a probe over the header that prints sizeof / offsetof
    of every field ctypes knows, and the numbers must agree."""
    import ctypes as C
    import subprocess
    from spokennlp_amd import lib
    structs = {"amdseg_bert_cfg": lib.BertCfg, "amdseg_bert_layer_params": lib.LayerParams, "amdseg_bert_layer_grads": lib.LayerGrads,
               "amdseg_bert_layer_acts": lib.LayerActs, "amdseg_bert_layer_ws": lib.LayerWs}
    hdr = open(os.path.join(ROOT, "include", "amdseg.h")).read()
    assert set(re.findall(r"typedef struct (amdseg_[a-z_]+) *\{", hdr)) == set(structs), "a struct of the header has no ctypes mirror"
    lines = ['#include <stdio.h>', '#include <stddef.h>', '#include "amdseg.h"', 'int main(void) {']
    for cname, cls in structs.items():
        lines.append(f'  printf("{cname} %zu\\n", sizeof({cname}));')
        for fname, _ in cls._fields_:
            lines.append(f'  printf("{cname}.{fname} %zu\\n", offsetof({cname}, {fname}));')
    lines += ['  return 0;', '}']
    src = tmp_path / "probe.c"
    src.write_text("\n".join(lines))
    exe = tmp_path / "probe"
    subprocess.run(["gcc", "-I", os.path.join(ROOT, "include"), str(src), "-o", str(exe)], check=True)
    got = dict(l.split() for l in subprocess.run([str(exe)], check=True, capture_output=True, text=True).stdout.splitlines())
    for cname, cls in structs.items():
        assert int(got[cname]) == C.sizeof(cls), cname
        for fname, _ in cls._fields_:
            assert int(got[f"{cname}.{fname}"]) == getattr(cls, fname).offset, f"{cname}.{fname}"


def test_bench_gpus_argument_launches_the_ranks(monkeypatch, capsys):
    """`python bench.py --gpus N` (the driver's command shape, no torchrun around it) must start N ranks itself -- VERDICT r04: args.gpus was
    parsed and never read, so the line silently measured dp1.  The argument path, on CPU: the launcher command, and the loud refusals."""
    import importlib
    import subprocess
    import sys
    import types
    bench = importlib.import_module("bench")
    seen = {}

    def fake_call(cmd, env=None):
        seen["cmd"], seen["env"] = cmd, env
        return 0
    monkeypatch.setattr(subprocess, "call", fake_call)
    monkeypatch.setenv("AMDSEG_DIST_BACKEND", "gloo")          # fewer GPUs than ranks here (none): allowed only over gloo
    monkeypatch.delenv("WORLD_SIZE", raising=False)
    rc = bench.self_launch(types.SimpleNamespace(gpus=4), ["--gpus", "4", "--steps", "2", "--warmup", "1"])
    assert rc == 0
    cmd = seen["cmd"]
    assert cmd[:3] == [sys.executable, "-m", "torch.distributed.run"] and "--nnodes=1" in cmd
    assert cmd[cmd.index("--nproc-per-node") + 1] == "4" and cmd[cmd.index("--master-addr") + 1] == "127.0.0.1"
    assert 1024 < int(cmd[cmd.index("--master-port") + 1]) < 65536
    tail = cmd[cmd.index("--master-port") + 2:]
    assert tail[0].endswith("bench.py") and tail[1:] == ["--gpus", "4", "--steps", "2", "--warmup", "1"]
    assert seen["env"]["HSA_ENABLE_IPC_MODE_LEGACY"] == "0"
    # without the gloo opt-in, more ranks than GPUs is refused, not silently run as dp1
    monkeypatch.delenv("AMDSEG_DIST_BACKEND")
    with pytest.raises(SystemExit) as e:
        bench.self_launch(types.SimpleNamespace(gpus=4), ["--gpus", "4"])
    assert "only 0 GPU(s) visible" in str(e.value)
    # main(): --gpus N without WORLD_SIZE goes to the launcher; a launcher that started another number of ranks is refused
    monkeypatch.setattr(sys, "argv", ["bench.py", "--gpus", "2", "--steps", "1", "--warmup", "0"])
    monkeypatch.setenv("AMDSEG_DIST_BACKEND", "gloo")
    with pytest.raises(SystemExit) as e:
        bench.main()
    assert e.value.code == 0 and seen["cmd"][seen["cmd"].index("--nproc-per-node") + 1] == "2"
    monkeypatch.setenv("WORLD_SIZE", "4")
    with pytest.raises(SystemExit) as e:
        bench.main()
    assert "must agree" in str(e.value)


def test_auto_registration_resolves_to_the_drop_in_classes(tmp_path):
    """the drop-in classes are registered with AutoModelForTokenClassification (config twins with their own model_type;
    transformers ignores register() for its native config classes) and the direct-class path is untouched.  Construction / save / load
    only -- no compute without a GPU."""
    import spokennlp_amd
    from spokennlp_amd import auto
    from transformers import AutoConfig, AutoModelForTokenClassification, BertConfig, ElectraConfig, LongformerConfig, BigBirdConfig
    small = dict(vocab_size=50, hidden_size=64, num_hidden_layers=1, num_attention_heads=1, intermediate_size=128, num_labels=2)
    stock = {"bert": BertConfig(**small), "electra": ElectraConfig(embedding_size=64, **small),
             "longformer": LongformerConfig(attention_window=[8], max_position_embeddings=70, type_vocab_size=1, pad_token_id=1, **small),
             "big_bird": BigBirdConfig(block_size=64, num_random_blocks=3, max_position_embeddings=1024, **small)}
    for fam, cfg in stock.items():
        ccls, mcls, direct = auto.FAMILIES[fam]
        twin = spokennlp_amd.amdseg_config(cfg, do_da_ts=True, cl_loss_weight=0.25)
        assert type(twin) is ccls and twin.model_type == "amdseg-" + fam.replace("_", "") and twin.hidden_size == 64 and twin.do_da_ts is True
        m = AutoModelForTokenClassification.from_config(twin)
        assert type(m) is mcls and isinstance(m, direct)
        assert set(m.state_dict()) == set(direct(cfg).state_dict())              # same HF parameter names on both paths
        d = tmp_path / fam
        m.save_pretrained(d)
        back = AutoConfig.from_pretrained(d)
        assert type(back) is ccls and back.cl_loss_weight == 0.25
        m2 = AutoModelForTokenClassification.from_pretrained(d)
        assert type(m2) is mcls and all(torch.equal(a, b) for a, b in zip(m.state_dict().values(), m2.state_dict().values()))
        # a stock config through Auto* stays stock (nothing built in is overridden)
        assert not isinstance(AutoModelForTokenClassification.from_config(cfg), direct)
    with pytest.raises(ValueError):
        from transformers import GPT2Config
        spokennlp_amd.amdseg_config(GPT2Config())


def test_product_sources_carry_no_probe_flags_and_few_switches():
    """the wrong-result timing probes (AMDSEG_ABL_*), the losing variants (merged attention backward, persistent NT form,
    fused bias + dropout + residual GEMM, LayerNorm pair forward) and their switches are out of the product; what a run can still be steered by
    from the environment fits on one screen."""
    import glob
    csrc = os.path.join(ROOT, "spokennlp_amd", "csrc")
    text = "".join(open(f).read() for f in glob.glob(os.path.join(csrc, "*.hip")) + glob.glob(os.path.join(csrc, "*.h")))
    for gone in ("AMDSEG_ABL_", "AMDSEG_PROBES", "AMDSEG_TN_A67", "EARLY_START", "attn_bwd_merged_kernel", "bias_drop_res", "PERSIST", "add_ln_fwd_pair768"):
        assert gone not in text, gone
    env_c = set(re.findall(r'getenv\("(AMDSEG_[A-Z0-9_]+)"\)', text))
    py = "".join(open(f).read() for f in glob.glob(os.path.join(ROOT, "spokennlp_amd", "*.py")))
    env_py = set(re.findall(r'environ(?:\.get|\.setdefault)?\(\s*"(AMDSEG_[A-Z0-9_]+)"', py)) | set(re.findall(r'environ\["(AMDSEG_[A-Z0-9_]+)"\]', py))
    switches = env_c | env_py
    assert len(switches) <= 15, sorted(switches)
    from spokennlp_amd import lib
    assert not os.path.exists(os.path.join(csrc, "attention_bwd_merged.hip"))
    assert "attention_bwd_merged.hip" not in open(os.path.join(ROOT, "spokennlp_amd", "build.py")).read()
    assert lib.ABI_VERSION == 14 and "amdseg_add_ln_fwd_keepmask" in lib.EXPORTS and "amdseg_ctx_create" in lib.EXPORTS and "amdseg_set_cu_budget" not in lib.EXPORTS


def test_erf_epilogue_build_flag_still_compiles(tmp_path):
    """-DAMDSEG_GELU_ERF_EPILOGUE (csrc/common.h: the exact erf form in the bf16 GELU epilogues instead of the fitted sigmoid form) is the one build
    flag the product sources keep: the deep-pipeline GEMM -- every GELU epilogue, including the one-byte derivative form -- must compile with it"""
    import shutil
    import subprocess
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(hipcc):
        pytest.skip("no hipcc on this box")
    csrc = os.path.join(ROOT, "spokennlp_amd", "csrc")
    r = subprocess.run([hipcc, "--offload-arch=gfx950", "-O1", "-std=c++17", "-fPIC", "-Wno-unused-result", "-DAMDSEG_GELU_ERF_EPILOGUE", "-c",
                        os.path.join(csrc, "gemm_dp.hip"), "-o", str(tmp_path / "gemm_dp_erf.o")], capture_output=True, text=True, cwd=csrc)
    assert r.returncode == 0, r.stderr[-1500:]
