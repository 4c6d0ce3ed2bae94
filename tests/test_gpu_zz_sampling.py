"""The sampler inside the captured decode step (example_generate.py:9-26).  In its own module, collected last: in
round 2 this test failed intermittently when it ran within the first seconds of the first GPU process on a fresh box
("Open at the end of round 2"); the assertions print what they compared."""
import pytest
import torch

pytestmark = pytest.mark.gpu


def test_sampling_step_matches_reference_sampler_semantics():
    """example_generate.py:9-26: temperature + top-k + exponential-race arg-max inside the captured step.
    top_k=1 is greedy; with top_k=5 every sampled token is one of the 5 largest logits of its own step,
    and the draws differ between steps (the graph-safe generator advances on replay)."""
    from quip_for_all_amd.decode import LlamaDecoder, SMALL as LLAMA_TINY
    dec = LlamaDecoder(LLAMA_TINY, max_len=64, device="cuda:0", seed=3)
    greedy = dec.generate(16, first_token=7)
    k1 = dec.generate(16, first_token=7, temperature=0.6, top_k=1)
    assert torch.equal(greedy, k1), (greedy.tolist(), k1.tolist())
    torch.manual_seed(0)
    dec.set_sampling(0.6, 5)
    dec.reset(7)
    dec.capture()
    dec.reset(7)
    toks, n_not_top1 = [], 0
    for _ in range(48):
        dec.graph.replay()
        lg = dec.step_logits.float()[0]
        t = int(dec.tok[0])
        top = torch.topk(lg, 5)
        assert lg[t] >= top.values[-1], (t, float(lg[t]), top.values.tolist())   # ties with the 5th logit stay candidates (logits < pivot are cut)
        n_not_top1 += bool(lg[t] < top.values[0])
        toks.append(t)
    assert n_not_top1 > 0, "48 draws at T=0.6 over 5 candidates never left the arg-max: sampler is not sampling"
    # back to greedy: re-captures and reproduces the greedy tokens
    again = dec.generate(16, first_token=7)
    assert torch.equal(again, greedy), (again.tolist(), greedy.tolist())
