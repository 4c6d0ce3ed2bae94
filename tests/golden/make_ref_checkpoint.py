#!/usr/bin/env python3
"""Write tiny checkpoints with the REFERENCE's own code and the logits its own loader + forward produce
(on-disk round trip against a real QuIP-for-all checkpoint).

    python tests/golden/make_ref_checkpoint.py          (authoring container, needs /root/reference)

For each variant a 2-layer Llama (hidden 256, ffn 688 -> exercises the K = 172 / K = 43 Hadamard factors) is
converted by the reference's `QuipQuantizer.convert_model`, its QuantLinear buffers are filled with seeded random
codes / signs / scales (there is no calibration data here; the on-disk format does not depend on the values),
saved by the reference's `QuipQuantizer.save` (accelerate's sharded `pytorch_model-0000i-of-0000N.bin` +
index + config.json + quantization_config.json), re-loaded by the reference's `load_quantized_model` and run on
fixed token ids.  Committed: the checkpoint directories and `ref_checkpoint_logits.npz` (data only).

    e8p12_norand      use_rand=False: had_left / had_right are NON-persistent buffers (qlinear.py:31-41), i.e.
                      absent from the checkpoint and rebuilt from the order-172 Hadamard table at load time
    e8p12_merge_suv   merge_suv=True: SU / SV dropped by QuantLinear.pack (qlinear.py:117-131), absent from the
                      checkpoint; K = 43 random orthogonal factors stored in the checkpoint

The reference only runs on CUDA; as in make_golden.py, CPU implementations of the three quip_lib ops it needs
are registered in this harness, built from the reference's own matmul_hadU and full-grid LUT, and
torch.cuda.is_available is patched for the duration of its loader's GPU check (quantizer.py:799-801)."""
import os
import shutil
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import make_golden  # noqa: E402

TOKENS = [[1, 5, 17, 99, 3, 64, 127, 8, 42, 77, 11, 100]]


def main():
    torch, quant, qlinear, codebook, e8p12 = make_golden._import_reference()
    import torch._custom_ops as co
    packed = e8p12.get_packed_abs_grid()
    full, _ = e8p12.get_full_grid(packed)
    full_f = full.float()

    def dec_e8p(Q):
        idx = Q.view(torch.int16).to(torch.int32) & 0xFFFF
        return full_f[idx.long()].reshape(Q.shape[0], -1).half()

    def had(x, scale):
        n = x.shape[-1]
        y = quant.matmul_hadU(x.float().reshape(-1, n), None, 1, n) * (n ** 0.5) * scale
        return y.reshape(x.shape).to(x.dtype)

    co.impl("quip_lib::hadamard", device_types="cpu")(had)
    co.impl("quip_lib::decompress_e8p_origorder", device_types="cpu")(lambda Q, g: dec_e8p(Q))
    co.impl("quip_lib::e8p_mm_origorder", device_types="cpu")(
        lambda x, Q, g: (x.float() @ dec_e8p(Q).float().T).to(x.dtype))

    import quantizer as refq
    from transformers import LlamaConfig, LlamaForCausalLM
    cfg = LlamaConfig(hidden_size=256, intermediate_size=688, num_hidden_layers=2, num_attention_heads=4,
                      num_key_value_heads=4, vocab_size=128, max_position_embeddings=64, tie_word_embeddings=False)
    logits = {}
    for name, kw in (("e8p12_norand", dict(use_rand=False, merge_suv=False)),
                     ("e8p12_merge_suv", dict(use_rand=True, merge_suv=True))):
        torch.manual_seed(1234)
        np.random.seed(1234)          # scipy's special_ortho_group (get_hadK, use_rand=True) draws from numpy's global state
        model = LlamaForCausalLM(cfg).half()
        qz = refq.QuipQuantizer(codebook="E8P12", ft_epochs=0, **kw)
        qz.convert_model(model)
        g = torch.Generator().manual_seed(99)
        for lname, layer in refq.get_layers(model, [qlinear.QuantLinear]).items():
            with torch.no_grad():
                layer.Qidxs.copy_(torch.randint(-32768, 32768, layer.Qidxs.shape, generator=g, dtype=torch.int32)
                                  .to(torch.int16))
                layer.Wscale.fill_(0.9 / (layer.in_features ** 0.5) / 1.2)
                if kw["merge_suv"]:          # what QuantLinear.pack does with merge_su / merge_sv (qlinear.py:125-131)
                    layer.SU = None
                    layer.SV = None
                else:
                    layer.SU.copy_((torch.randint(0, 2, layer.SU.shape, generator=g) * 2 - 1).half())
                    layer.SV.copy_((torch.randint(0, 2, layer.SV.shape, generator=g) * 2 - 1).half())
        out = os.path.join(HERE, "ref_checkpoint_" + name)
        shutil.rmtree(out, ignore_errors=True)
        qz.save(model, out, max_shard_size="200KB")        # the reference's own writer
        del model
        real = torch.cuda.is_available
        torch.cuda.is_available = lambda: True                # quantizer.py:799-801; everything below stays on the CPU
        try:
            loaded = refq.load_quantized_model(out, device_map={"": "cpu"})
        finally:
            torch.cuda.is_available = real
        with torch.no_grad():
            lg = loaded(torch.tensor(TOKENS)).logits
        logits[name] = lg.float().numpy()
        print(name, sorted(os.listdir(out)), "logits", lg.shape, float(lg.abs().max()))
    np.savez_compressed(os.path.join(HERE, "ref_checkpoint_logits.npz"), tokens=np.asarray(TOKENS, dtype=np.int64), **logits)


if __name__ == "__main__":
    main()
