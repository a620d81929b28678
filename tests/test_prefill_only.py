"""The prefill-only workflow.  CPU: the scheduler under the three scenarios of the reference's own
test (tests/prefill_only/test_scheduler.py:27-129: limited by the request budget, limited by the
token budget, prompts over max_model_len ignored) with the same parameters; config checks; aborts.
GPU: the encoder engine against a plain fp32 torch forward; sync == async; replicas."""
import pytest
import torch

import light_vllm_amd  # noqa: F401
from light_vllm_amd.prefill_only import (PrefillOnlyRequestOutput, PrefillOnlyScheduler,
                                         PrefillOnlySchedulerConfig as SchedulerConfig, Request,
                                         SchedulableRequest)


def processor(num_new_tokens):
    return lambda r: SchedulableRequest(request_id=r.request_id, arrival_time=r.arrival_time,
                                        prompt_token_ids=[0] * num_new_tokens)


def finished(out):
    return [PrefillOnlyRequestOutput(r.request_id, None, r.prompt_token_ids, True) for r in out.scheduled_requests]


@pytest.mark.parametrize("num_new_tokens", [9, 99, 199])
@pytest.mark.parametrize("n_request", [9, 99, 199])
@pytest.mark.parametrize("max_num_requests", [1, 2, 3, 5, 7])
def test_limited_by_max_num_requests(n_request, num_new_tokens, max_num_requests):
    max_model_len = num_new_tokens + 1
    s = PrefillOnlyScheduler(SchedulerConfig(max_num_batched_tokens=max_model_len * max_num_requests,
                                             max_model_len=max_model_len, max_num_seqs=max_num_requests),
                             processor(num_new_tokens))
    for i in range(1, n_request + 1):
        s.add_request(Request(request_id=str(i), arrival_time=0.0))
    while s.has_unfinished_requests():
        out = s.schedule()
        s.free_finished_request(finished(out))
        if s.has_unfinished_requests():
            assert len(out.scheduled_requests) == max_num_requests
        else:
            assert len(out.scheduled_requests) <= max_num_requests
        assert len(out.ignored_requests) == 0


@pytest.mark.parametrize("num_new_tokens", [9, 99, 199])
@pytest.mark.parametrize("n_request", [9, 99, 199])
@pytest.mark.parametrize("max_num_requests", [2, 3, 5, 7])
def test_limited_by_token_budget(n_request, num_new_tokens, max_num_requests):
    s = PrefillOnlyScheduler(SchedulerConfig(max_model_len=num_new_tokens + 1, max_num_seqs=max_num_requests,
                                             max_num_batched_tokens=(num_new_tokens + 1) * (max_num_requests - 1)),
                             processor(num_new_tokens))
    for i in range(1, n_request + 1):
        s.add_request(Request(request_id=str(i), arrival_time=0.0))
    n = 0
    while s.has_unfinished_requests():
        out = s.schedule()
        n += len(out.scheduled_requests)
        s.free_finished_request(finished(out))
        if s.has_unfinished_requests():
            assert len(out.scheduled_requests) == max_num_requests - 1
        else:
            assert len(out.scheduled_requests) <= max_num_requests - 1
        assert len(out.ignored_requests) == 0
    assert n == n_request


@pytest.mark.parametrize("num_new_tokens", [9, 99, 199])
@pytest.mark.parametrize("n_request", [9, 99, 199])
@pytest.mark.parametrize("max_num_requests", [2, 3, 5, 7])
def test_ignored_requests(n_request, num_new_tokens, max_num_requests):
    max_model_len = num_new_tokens // 2
    s = PrefillOnlyScheduler(SchedulerConfig(max_num_batched_tokens=max_model_len * max_num_requests,
                                             max_model_len=max_model_len, max_num_seqs=max_num_requests),
                             processor(num_new_tokens))
    for i in range(1, n_request + 1):
        s.add_request(Request(request_id=str(i), arrival_time=0.0))
    n_ignored = 0
    while s.has_unfinished_requests():
        out = s.schedule()
        assert len(out.scheduled_requests) == 0 and len(out.ignored_requests) > 0
        n_ignored += len(out.ignored_requests)
    assert n_ignored == n_request


def test_config_checks_and_aborts():
    with pytest.raises(ValueError, match="max_num_batched_tokens"):
        SchedulerConfig(max_model_len=100, max_num_batched_tokens=50, max_num_seqs=4)
    with pytest.raises(ValueError, match="max_num_on_the_fly"):
        SchedulerConfig(max_model_len=10, max_num_seqs=4, max_num_on_the_fly=1)
    with pytest.raises(ValueError, match="scheduling"):
        SchedulerConfig(max_model_len=10, max_num_seqs=4, scheduling="bogus")
    assert SchedulerConfig(max_model_len=10, max_num_seqs=4, scheduling="double_buffer").max_num_on_the_fly == 3
    cfg = SchedulerConfig(max_model_len=10, max_num_seqs=4)
    assert cfg.max_num_batched_tokens == 40 and cfg.max_num_on_the_fly == 2
    s = PrefillOnlyScheduler(cfg, processor(5))
    for i in range(6):
        s.add_request(Request(str(i)))
    s.add_request(Request("3"))  # duplicate id: ignored
    s.abort_request(["1", "4"])
    out = s.schedule()
    assert [r.request_id for r in out.scheduled_requests] == ["0", "2", "3", "5"]
    outs = s.remove_abort_request(finished(out))
    s.free_finished_request(outs)
    assert not s.has_unfinished_requests()


# ------------------------------------------------------------------ GPU: the encoder engine
DEV = "cuda:0"


def reference_hidden(model, token_ids):
    """fp32 torch forward of EncoderModel's weights on one sequence, dense bidirectional attention."""
    import torch.nn.functional as F
    cfg = model.cfg
    hid, H, D = cfg.hidden_size, cfg.num_attention_heads, cfg.head_dim
    ids = torch.tensor(token_ids, device=model.device)
    T = ids.numel()
    pos = torch.arange(T, device=model.device) + cfg.pad_token_id + 1
    f = lambda t: t.float()
    x = f(model.word_emb)[ids] + f(model.pos_emb)[pos] + f(model.type_emb)[0]
    x = F.layer_norm(x, (hid,), f(model.emb_ln[0]), f(model.emb_ln[1]), cfg.layer_norm_eps)
    for lw in model.layers:
        qkv = x @ f(lw.qkv_w).T + f(lw.qkv_b)
        q, k, v = (t.view(T, H, D).transpose(0, 1) for t in qkv.split([hid, hid, hid], dim=-1))
        att = torch.softmax(q @ k.transpose(1, 2) / D ** 0.5, dim=-1) @ v
        a = att.transpose(0, 1).reshape(T, hid)
        x = F.layer_norm(x + a @ f(lw.out_w).T + f(lw.out_b), (hid,), f(lw.attn_ln[0]), f(lw.attn_ln[1]), cfg.layer_norm_eps)
        h = F.gelu(x @ f(lw.fc1_w).T + f(lw.fc1_b))
        x = F.layer_norm(x + h @ f(lw.fc2_w).T + f(lw.fc2_b), (hid,), f(lw.out_ln[0]), f(lw.out_ln[1]), cfg.layer_norm_eps)
    return x


def make_encoder(scheduling="sync", pooling="cls", max_seqs=4, budget=None):
    from light_vllm_amd.prefill_only.engine import PrefillOnlyEngine
    from light_vllm_amd.prefill_only.model import EncoderConfig
    return PrefillOnlyEngine(EncoderConfig.tiny(), SchedulerConfig(max_model_len=512, max_num_seqs=max_seqs,
                                                                   max_num_batched_tokens=budget, scheduling=scheduling),
                             device=DEV, pooling=pooling, seed=0)


def encoder_prompts():
    g = torch.Generator().manual_seed(1)
    return [torch.randint(2, 512, (n,), generator=g).tolist() for n in (1, 5, 17, 64, 33, 100, 256, 2, 31, 129)]


@pytest.mark.gpu
def test_encoder_engine_matches_fp32_reference():
    eng = make_encoder(pooling="last_hidden_states")
    ps = encoder_prompts()
    res = eng.encode(ps, use_async=False)
    for i, p in enumerate(ps):
        want = reference_hidden(eng.model, p).cpu()
        got = res[str(i)].float()
        assert got.shape == want.shape
        cos = torch.nn.functional.cosine_similarity(got, want, dim=1)
        assert float(cos.min()) >= 0.999, (i, float(cos.min()))
        assert float((got - want).abs().max()) <= 3e-2 * float(want.abs().max())


@pytest.mark.gpu
@pytest.mark.parametrize("pooling", ["cls", "mean"])
def test_encoder_async_equals_sync_and_pooling(pooling):
    ps = encoder_prompts() + [[7] * 600]  # the last one exceeds max_model_len: ignored
    sync = make_encoder("sync", pooling).encode(ps, use_async=False)
    e = make_encoder("async", pooling, max_seqs=3)
    asyn = e.encode(ps, use_async=True)
    e.shutdown()
    assert sync[str(len(ps) - 1)] is None and asyn[str(len(ps) - 1)] is None
    for i in range(len(ps) - 1):
        a, b = sync[str(i)], asyn[str(i)]
        assert a.shape == (128,) and abs(float(a.norm()) - 1.0) < 1e-3
        # batch composition differs between the two runs; per-sequence attention does not depend on it
        assert float((a - b).abs().max()) <= 2e-2
    ref = make_encoder(pooling="last_hidden_states")
    h = ref.encode(ps[:3], use_async=False)
    for i in range(3):
        x = h[str(i)].float()
        want = torch.nn.functional.normalize(x[0] if pooling == "cls" else x.mean(0), dim=-1)
        assert float((sync[str(i)] - want).abs().max()) <= 2e-2
