"""GPU: smoke of tools/mem_spd.py, the mem_spd_test.py-shaped harness (prefill, then decode steps across a 256-token
trigger, 1 warm-up + timed repeats, ms per generate + peak memory): small batch / few layers, the three call sequences
and the graph-replayed fused form must end in the same attention output (fp16: rtol 4e-3, atol 2e-3)."""
import importlib.util
import os

import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _load():
    spec = importlib.util.spec_from_file_location("mem_spd", os.path.join(ROOT, "tools", "mem_spd.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def test_mem_spd_harness_smoke():
    ms = _load()
    # prompt 500 -> 256 compressed + 244 in the window: the trigger fires at decode step 44 of 50
    common = ["--batch", "2", "--layers", "2", "--prompt-length", "500", "--output-length", "50", "--repeats", "1"]
    res, outs = ms.main(["--api", "fused", "native", "reference"] + common)
    assert [r["api"] for r in res] == ["fused", "native", "reference"]
    for r in res:
        assert r["triggers_per_generate"] == 1 and r["final_compressed_tokens"] == 512 and r["final_kv_seq_len"] == 550
        assert r["ms_per_generate_avg"] > 0 and r["peak_mem_gb"] > 0 and len(r["ms_per_generate"]) == 1
    torch.testing.assert_close(outs["fused"].float(), outs["native"].float(), rtol=4e-3, atol=2e-3)
    torch.testing.assert_close(outs["reference"].float(), outs["native"].float(), rtol=4e-3, atol=2e-3)
    res_g, outs_g = ms.main(["--api", "fused", "--graph"] + common)
    assert res_g[0]["api"] == "fused+graph" and res_g[0]["triggers_per_generate"] == 1 and res_g[0]["final_kv_seq_len"] == 550
    torch.testing.assert_close(outs_g["fused"].float(), outs["native"].float(), rtol=4e-3, atol=2e-3)


def test_checkpoint_must_be_a_local_directory():
    ms = _load()
    with pytest.raises(SystemExit, match="local directory"):
        ms.main(["--checkpoint", "meta-llama/Meta-Llama-3-8B-Instruct"])
