"""Pipeline-level behaviour on the device: the k1 / k2 clamp quirk (Hybridgl_main.py:178-181) with images that yield
fewer than 3 / 6 proposals, grouped vs ref-by-ref steps, the overlapped software pipeline, the empty-proposal path and
the two-rank benchmark (bench.py --gpus 2) against the single-rank run over the same refs."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _tiny(cuda, **kw):
    from hybridgl_amd import weights
    from hybridgl_amd.backbone import CLIPViTFM
    from hybridgl_amd.pipeline import HybridGLPipeline
    model = CLIPViTFM("tiny", state_dict=weights.clip_state_dict("tiny", 0), device=cuda)
    return HybridGLPipeline(model, masking_block=9, res=64, **kw)


def _ref(i, cuda, N):
    from hybridgl_amd.pipeline import synthetic_ref
    return synthetic_ref(i, cuda, N=N, H=96, W=128, context=16, vocab=512)


def _oracle_indices(pipe, ref, host, k1, k2):
    """winning indices of every sentence of `ref` from the numpy oracle, fed with the device's hybrid / text features"""
    from hybridgl_amd.pipeline import black_for
    from oracle import clip_oracle as O
    loc, glo = O.synthesize_views(host["img"], host["blur"], host["norm"], host["masks"], 64)
    hybrid = pipe.model(torch.from_numpy(loc).to(pipe.model.device), torch.from_numpy(glo).to(pipe.model.device), ref.masks,
                        masking_block=9, fusion_mode=pipe.fusion_mode).cpu().numpy()
    text = pipe.model.model.encode_text(ref.tokens).cpu().numpy()
    out = []
    for j, s in enumerate(ref.sentences):
        gem = O.coherence_scores(host["attn"][j], host["masks"], s.dirflag, black_for(s.relaflag))
        ens = 0.5 * text[s.sentence_row] + 0.5 * text[s.noun_phrase_row]
        ip, ifin, _, _ = O.score_sentence(hybrid, ens, text[s.other_noun_rows].mean(0), host["boxes"], gem, 100.0, k1, k2, 0.6,
                                          s.relaflag, s.n_nouns != 0)
        out.append((ip, ifin))
    return out


@pytest.mark.parametrize("k_clamp", ["persistent", "per_ref"])
def test_k_clamp_with_few_proposals(cuda, k_clamp):
    """An image with 5 proposals clamps k2 to 5, one with 2 clamps k1 and k2 to 2; the reference never restores them
    (persistent).  Every sentence's winners are checked against the oracle run with the k1 / k2 each mode prescribes."""
    pipe = _tiny(cuda, k_clamp=k_clamp)
    plan = [(0, 12), (1, 5), (2, 12), (3, 2), (4, 12)]          # (ref seed, number of proposals)
    want_k = {"persistent": [(3, 6), (3, 5), (3, 5), (2, 2), (2, 2)],
              "per_ref": [(3, 6), (3, 5), (3, 6), (2, 2), (3, 6)]}[k_clamp]
    for (i, n), (k1, k2) in zip(plan, want_k):
        ref, host = _ref(i, cuda, n)
        host["blur"] = host["blur"]
        n0 = len(pipe.iu_log)
        _, _, last = pipe.step(ref)
        assert (pipe.k1, pipe.k2) == (k1, k2)
        want = _oracle_indices(pipe, ref, host, k1, k2)
        rows = pipe.partial_rows()[n0:]
        from oracle import clip_oracle as O
        for (ip, ifin), row in zip(want, rows):
            assert [int(v) for v in row[2:4]] == list(O.compute_iou(host["masks"][ip], host["gt"]))
            assert [int(v) for v in row[4:6]] == list(O.compute_iou(host["masks"][ifin], host["gt"]))
        assert [int(v) for v in last[0].cpu()] == list(want[-1])


def test_persistent_clamp_sequence_vs_reference_golden(cuda, golden_dir):
    """tests/golden/scoring_small.npz: the reference's own tail run over refs with 12, 5, 12, 2, 12, 2 proposals with the
    k1 / k2 it carries from ref to ref.  The pipeline's scoring stage (k_clamp="persistent") must pick the same masks,
    count the same I / U and end every ref with the same k1 / k2."""
    from hybridgl_amd.pipeline import RefBatch, Sentence
    g = np.load(os.path.join(golden_dir, "scoring_small.npz"))
    from oracle.cases import tail_case
    pipe = _tiny(cuda, k_clamp="persistent")
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(cuda)
    for step, rec in enumerate(g["plan"]):
        ci, N, rela, dirflag, has_other = str(rec).split(",")
        ci, N, has_other = int(ci), int(N), bool(int(has_other))
        hybrid, t_pos, t_neg, masks, boxes, attn, gt = tail_case(ci, N, 32, 96, 128)
        text = t(np.concatenate([t_pos, t_pos, t_neg], axis=0))       # sentence == noun phrase makes the r-ensemble exact
        sent = Sentence(0, 1, [2], dirflag, rela, 1 if has_other else 0, t(attn))
        ref = RefBatch(None, None, None, t(masks), t(boxes), None, t(gt), [sent], index=step)
        idx = pipe._score_ref(ref, t(hybrid), text, None)[0]
        assert [int(v) for v in idx.cpu()] == [int(v) for v in g[f"s{step}_idx"]], step
        assert [pipe.k1, pipe.k2] == [int(v) for v in g[f"s{step}_k"]], step
        assert [int(v) for v in pipe.partial_rows()[-1][4:6]] == [int(v) for v in g[f"s{step}_IU"]], step


def test_exact_score_ties_vs_reference_golden(cuda, golden_dir):
    """tests/golden/scoring_ties.npz: refs whose proposals contain exact copies (identical feature rows, masks, boxes) --
    the scores tie exactly, and the device's arg-max / top-k must break the ties the way torch's do in the reference
    (Hybridgl_main.py:163-183): same winning indices, same I / U."""
    from hybridgl_amd.pipeline import RefBatch, Sentence
    from oracle.cases import TIE_PLAN, tie_case
    g = np.load(os.path.join(golden_dir, "scoring_ties.npz"))
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(cuda)
    for step, (ci, dup, rela, dirflag, has_other) in enumerate(TIE_PLAN):
        pipe = _tiny(cuda, k_clamp="per_ref")
        hybrid, t_pos, t_neg, masks, boxes, attn, gt = tie_case(ci, dup)
        text = t(np.concatenate([t_pos, t_pos, t_neg], axis=0))
        sent = Sentence(0, 1, [2], dirflag, rela, 1 if has_other else 0, t(attn))
        ref = RefBatch(None, None, None, t(masks), t(boxes), None, t(gt), [sent], index=step)
        idx = pipe._score_ref(ref, t(hybrid), text, None)[0]
        assert [int(v) for v in idx.cpu()] == [int(v) for v in g[f"t{step}_idx"]], (step, dup)
        assert [int(v) for v in pipe.partial_rows()[-1][4:6]] == [int(v) for v in g[f"t{step}_IU"]], step


def test_text_glue_vs_reference_golden(cuda, golden_dir):
    """tests/golden/tail_glue.npz: sentence / noun phrase / 0..3 other nouns as TOKENS; the reference encodes them, mixes
    r * sentence + (1 - r) * noun phrase, averages the other nouns (Hybridgl_main.py:146-165) and runs its tail.  The
    device text encoder + score_sentence_kernel (which does the mix and the mean itself) must pick the same masks and
    agree on the soft-maxed scores."""
    from hybridgl_amd.pipeline import RefBatch, Sentence
    from oracle.cases import GLUE_PLAN, glue_tokens, tail_case
    g = np.load(os.path.join(golden_dir, "tail_glue.npz"))
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(cuda)
    for ci, n_other, rela, dirflag in GLUE_PLAN:
        pipe = _tiny(cuda, k_clamp="per_ref", r=float(g["r"][0]))
        hybrid, _, _, masks, boxes, attn, gt = tail_case(ci, 12, 32, 96, 128)
        text = pipe.model.model.encode_text(t(glue_tokens(ci, n_other)))
        sent = Sentence(0, 1, list(range(2, 2 + n_other)), dirflag, rela, n_other, t(attn))
        ref = RefBatch(None, None, None, t(masks), t(boxes), None, t(gt), [sent], index=ci)
        idx, sc, _, _ = pipe._score_ref(ref, t(hybrid), text, None)
        assert [int(v) for v in idx.cpu()] == [int(v) for v in g[f"g{ci}_idx"]], ci
        np.testing.assert_allclose(torch.softmax(sc.reshape(-1), 0).cpu().numpy(), g[f"g{ci}_score_clip"][:, 0], rtol=0, atol=2e-6)   # sc: logits
        assert [int(v) for v in pipe.partial_rows()[-1][4:6]] == [int(v) for v in g[f"g{ci}_IU"]], ci


def test_divisions_by_zero_vs_reference_golden(cuda, golden_dir):
    """tests/golden/scoring_nan.npz: constant heat-map, empty / full proposal masks (also as the best-scoring proposal): the
    device's fused coherence + scoring kernels must report the indices the reference reports when NaNs reach its arg-max"""
    from hybridgl_amd.pipeline import RefBatch, Sentence
    from oracle.cases import NAN_PLAN, nan_case
    g = np.load(os.path.join(golden_dir, "scoring_nan.npz"))
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(cuda)
    for step, (kind, rela, dirflag, has_other) in enumerate(NAN_PLAN):
        pipe = _tiny(cuda, k_clamp="per_ref")
        hybrid, t_pos, t_neg, masks, boxes, attn, gt = nan_case(kind)
        text = t(np.concatenate([t_pos, t_pos, t_neg], axis=0))
        sent = Sentence(0, 1, [2], dirflag, rela, 1 if has_other else 0, t(attn))
        ref = RefBatch(None, None, None, t(masks), t(boxes), None, t(gt), [sent], index=step)
        idx = pipe._score_ref(ref, t(hybrid), text, None)[0]
        if kind.startswith("nan_row"):      # NaN features: only the pure-CLIP index is defined (oracle/cases.py)
            assert int(idx.cpu()[0]) == int(g[f"n{step}_idx"][0]), (step, kind)
            continue
        assert [int(v) for v in idx.cpu()] == [int(v) for v in g[f"n{step}_idx"]], (step, kind)
        assert [int(v) for v in pipe.partial_rows()[-1][4:6]] == [int(v) for v in g[f"n{step}_IU"]], (step, kind)


def test_clamp_is_per_rank_under_sharding(cuda):
    """Documented deviation: with k_clamp="persistent" the quirk acts on the items of ONE process in its
    own order.  Two 'ranks' that split [12, 2, 12, 12] proposals as (0, 2) / (1, 3) clamp only rank 1's later item; the
    single process clamps items 2 and 3.  k_clamp="per_ref" is the same on any split."""
    counts = [12, 2, 12, 12]
    def run(indices, mode):
        pipe = _tiny(cuda, k_clamp=mode)
        ks = {}
        for i in indices:
            pipe.step(_ref(i, cuda, counts[i])[0])
            ks[i] = (pipe.k1, pipe.k2)
        return ks, pipe.partial_rows()
    single, rows1 = run(range(4), "persistent")
    assert single == {0: (3, 6), 1: (2, 2), 2: (2, 2), 3: (2, 2)}
    r0, _ = run([0, 2], "persistent")
    r1, _ = run([1, 3], "persistent")
    assert r0 == {0: (3, 6), 2: (3, 6)} and r1 == {1: (2, 2), 3: (2, 2)}       # item 2 differs from the single process
    from hybridgl_amd import dist as D
    _, a = run([0, 2], "per_ref")
    _, b = run([1, 3], "per_ref")
    _, whole = run(range(4), "per_ref")
    assert D.metrics_from_rows(np.concatenate([a, b])) == D.metrics_from_rows(whole)


def test_grouped_run_equals_ref_by_ref(cuda):
    """run() on given proposals (one text batch + one hybrid forward for 8 refs) and the metric rows it files are identical
    to eight step() calls: every mask row and every string is independent of its batch."""
    refs = [_ref(i, cuda, 12)[0] for i in range(8)]
    a, b = _tiny(cuda), _tiny(cuda)
    for r in refs:
        a.step(r)
    assert b.run(iter(refs), group=8, collect=True) == 8
    outs = b.collected
    assert len(outs) == 8
    assert np.array_equal(a.partial_rows(), b.partial_rows())
    assert a.metrics() == b.metrics()
    for r, (hyb, text, last) in zip(refs, outs):
        # the row count decides which GEMM kernel runs (skinny / fp32 below 512 rows at this tiny geometry), so the features
        # agree to rounding, the winners exactly
        h1, t1, l1 = _tiny(cuda).step(r)
        assert torch.allclose(h1, hyb, atol=2e-5, rtol=0) and torch.allclose(t1, text, atol=2e-5, rtol=0)
        assert torch.equal(l1[0], last[0])


def test_two_stream_run_with_discarded_proposals_equals_serial(cuda):
    """--proposals-from seeded (rounds 1-2's benchmark workload): the SAM proposal kernels of group g+1 run beside the CLIP
    stage of group g and their output is discarded; the rows are those of plain step() calls on the same refs (ViT-B/16 +
    SAM at the tiny geometry), on two streams and on one."""
    from hybridgl_amd.backbone import CLIPViTFM
    from hybridgl_amd.pipeline import HybridGLPipeline, synthetic_ref
    from hybridgl_amd.sam import SamAutomaticMaskGenerator, sam_model_registry
    model = CLIPViTFM("ViT-B/16", seed=0, device=cuda)
    gen = SamAutomaticMaskGenerator(sam_model_registry["tiny"](device=cuda), points_per_side=4, pred_iou_thresh=-1e30,
                                    stability_score_thresh=0.0, box_nms_thresh=2.0, min_mask_region_area=50)
    refs = [synthetic_ref(i, cuda, N=8, H=160, W=200, sam_img_size=256)[0] for i in range(4)]
    mk = lambda: HybridGLPipeline(model, mask_generator=gen, use_sam_masks=False, cleanup_given_masks=True)
    a, b, c = mk(), mk(), mk()
    for r in refs:
        a.step(r)
    b.run(iter(refs), group=2)
    c.run(iter(refs), group=2, serial=True)
    torch.cuda.synchronize()
    assert np.array_equal(a.partial_rows(), b.partial_rows()) and np.array_equal(a.partial_rows(), c.partial_rows())
    assert b.mask_generator is gen          # the generator is never swapped out to steer the step


def test_all_empty_masks_and_empty_proposals(cuda):
    """All-empty proposal masks still score (scores are finite, IoU = 0 / |gt|); a proposal stage that keeps nothing
    raises EmptyProposals (hybridgl_amd.main counts and skips such refs)."""
    import dataclasses
    from hybridgl_amd.pipeline import EmptyProposals
    pipe = _tiny(cuda)
    ref, host = _ref(0, cuda, 12)
    ref = dataclasses.replace(ref, masks=torch.zeros_like(ref.masks))
    hyb, _, last = pipe.step(ref)
    assert torch.isfinite(hyb).all() and torch.isfinite(last[1]).all()
    rows = pipe.partial_rows()
    assert (rows[:, 2] == 0).all() and (rows[:, 3] == int(host["gt"].sum())).all()

    class NoMasks:
        crop_n_layers = 0
        def generate_device(self, img, resized=None, fixed_n=None):
            z = torch.zeros((0,) + tuple(img.shape[:2]), dtype=torch.uint8, device=img.device)
            return z, torch.zeros((0, 4), dtype=torch.int64, device=img.device), None, None
    p2 = _tiny(cuda, mask_generator=NoMasks(), use_sam_masks=True)
    with pytest.raises(EmptyProposals):
        p2.step(_ref(1, cuda, 12)[0])


def _bench(args, timeout=1500):
    env = dict(os.environ)
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args, env=env, capture_output=True, text=True,
                       timeout=timeout)
    assert r.returncode == 0, r.stderr[-3000:]
    line = [l for l in r.stdout.splitlines() if l.startswith("{")][-1]
    return json.loads(line)


def test_bench_two_ranks(cuda):
    """`python bench.py --gpus 2` (no launcher) starts two ranks, which here share the GPU (metric exchange over gloo);
    the report covers 2 x steps refs and equals the single-rank run over the same 16 refs, row for row."""
    common = ["--scope", "A", "--warmup", "0", "--no-cpu-baseline", "--no-also", "--masks", "16"]
    two = _bench(["--gpus", "2", "--steps", "8", "--pool", "8"] + common)
    one = _bench(["--gpus", "1", "--steps", "16", "--pool", "16"] + common)
    # two ranks on ONE device: n_gpus is the hardware (1), the ranks show in world_size_seen / ranks_per_gpu
    assert two["n_gpus"] == 1 and two["ranks_per_gpu"] == 2 and two["world_size_seen"] == 2 and one["n_gpus"] == 1
    assert two["metrics"]["n_sentences"] == one["metrics"]["n_sentences"] == 48
    assert two["metrics"] == one["metrics"]
    assert two["value"] > 0 and two["scaling"] == "weak"


def test_bench_two_ranks_at_the_drivers_arguments(cuda):
    """The driver's multi-GPU command shape -- `bench.py --gpus 2 --steps 20 --warmup 5`, full scope-B workload -- as two ranks
    that share this box's one GPU (gloo): rank 0's line reports a world of two, the metric rows of BOTH ranks (2 x 20 refs x 3
    sentences), the first-use set-up (HybridGLPipeline.prepare) on every rank, and a whole-job rate within 15 % of the one-rank
    run of the same command: two ranks on one device can only share it, so a first-use or rendezvous cost paid inside the
    timed region of either rank would show here exactly as it would on eight GPUs."""
    quick = ["--steps", "20", "--warmup", "5", "--no-cpu-baseline", "--no-also", "--no-live-pmc", "--no-disk", "--no-rccl-check"]
    two = _bench(["--gpus", "2"] + quick)
    one = _bench(["--gpus", "1"] + quick)
    assert two["world_size_seen"] == 2 and two["ranks_per_gpu"] == 2 and two["n_gpus"] == 1 and two["backend"].startswith("gloo")
    assert two["steps"] == 20 and two["warmup"] == 5 and two["scaling"] == "weak"
    assert two["metrics"]["n_sentences"] == 2 * 20 * 3 and one["metrics"]["n_sentences"] == 20 * 3
    assert two["config"]["prepared"] is not None and one["config"]["prepared"] is not None
    assert two["split_overflow_count"] == 0
    # whole job: 40 refs in max-over-ranks time; the device is shared, so the total rate is that of one rank
    assert two["value"] >= 0.85 * one["value"], (two["value"], one["value"], two["timed_region"], one["timed_region"])
    # and the one-rank run itself is at its steady state: no first-use cost left in the 20 timed steps
    # (the allocator may still round a request up past every cached block once in a while: two device mallocs at most, where an
    # unprepared run makes six and grows its reservation by 20 GiB inside the timed region)
    assert one["timed_region"]["device_mallocs"] <= 2 and two["timed_region"]["device_mallocs"] <= 2, (one["timed_region"], two["timed_region"])


def test_rccl_backend_at_world_size_one(cuda):
    """The `nccl` (= RCCL) branch of hybridgl_amd/dist.py -- init_process_group with device_id, all_gather / all_reduce on
    DEVICE tensors -- executed on the 1-GPU box in a world of one (a child process with a time limit: a communicator that
    cannot come up must fail this test, not hang the suite)."""
    import subprocess
    import sys
    code = ("import json, sys, torch; sys.path.insert(0, %r); from hybridgl_amd import dist as D; "
            "torch.cuda.set_device(0); print('HGL_SELFCHECK', json.dumps(D.rccl_selfcheck(torch.device('cuda', 0))))" % ROOT)
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    line = [l for l in r.stdout.splitlines() if l.startswith("HGL_SELFCHECK ")][-1]     # RCCL prints its own banner to stdout
    got = json.loads(line[len("HGL_SELFCHECK "):])
    assert got["backend"] == "nccl" and got["world_size_seen"] == 1 and got["rows_roundtrip_ok"] and got["max_ok"], got


@pytest.mark.parametrize("N,H,W,n_sent", [(64, 640, 640, 3), (13, 97, 131, 5), (7, 120, 160, 9), (30, 200, 150, 18), (1, 64, 64, 2)])
def test_fused_tail_equals_per_sentence_launches(cuda, N, H, W, n_sent):
    """hgl_score_ref (one call per ref: every mask byte read once for all sentences' heat-maps, one scoring workgroup per
    sentence, both IoUs and the accumulators in the same four launches) against hgl_coherence_scores + hgl_score_sentence +
    2 x hgl_iou_select per sentence: indices, counts, accumulators, logits and coherence scores BIT FOR BIT -- ragged sizes,
    more sentences than one pooling pass (4) and than one launch (16) hold, a single proposal (k1 / k2 clamp)."""
    from hybridgl_amd.backbone import CLIPViTFM
    from hybridgl_amd.pipeline import HybridGLPipeline, synthetic_ref
    model = CLIPViTFM("tiny", seed=0, device=cuda)
    refs = [synthetic_ref(i, cuda, N=N, H=H, W=W, n_sent=n_sent, vocab=512, context=16)[0] for i in range(2)]
    for r in refs:        # a ref whose second sentence has no other noun and whose last has two (consecutive rows)
        r.sentences[1].other_noun_rows = []
        r.sentences[1].n_nouns = 0
    a, b = HybridGLPipeline(model, res=64), HybridGLPipeline(model, res=64)
    a.fused_tail, b.fused_tail = True, False
    outs = []
    for p in (a, b):
        outs.append([p.step(r)[2] for r in refs])
    torch.cuda.synchronize()
    assert np.array_equal(a.partial_rows(), b.partial_rows()) and a.partial_rows().shape == (2 * n_sent, 6)
    assert np.array_equal(a.winning_indices(), b.winning_indices())
    assert torch.equal(a.cum, b.cum) and int(a.cum[1]) > 0
    for x, y in zip(*outs):
        for u, v in zip(x, y):
            assert torch.equal(u, v, ) or (torch.isnan(u) == torch.isnan(v)).all() and torch.equal(torch.nan_to_num(u), torch.nan_to_num(v))


def test_group_tail_equals_per_ref_tail(cuda):
    """hgl_score_group (the tails of ALL refs of a group in one set of four launches, a per-ref descriptor table in device
    memory, grids sized by the largest ref) against hgl_score_ref per ref through the grouped loop run(): rows, winning
    indices, accumulators and the collected last-sentence tensors BIT FOR BIT -- refs of different image sizes, proposal
    counts and sentence counts in one group, a ref with a single proposal (k clamp), more refs than one launch holds (16),
    the persistent clamp of Hybridgl_main.py:178-181 carried from ref to ref."""
    from hybridgl_amd.backbone import CLIPViTFM
    from hybridgl_amd.pipeline import HybridGLPipeline, synthetic_ref
    model = CLIPViTFM("tiny", seed=0, device=cuda)
    shapes = [(13, 97, 131, 5), (7, 120, 160, 3), (30, 200, 150, 9), (1, 64, 64, 2), (5, 80, 72, 16)]
    refs = []
    for i in range(19):
        N, H, W, n_sent = shapes[i % len(shapes)]
        r = synthetic_ref(i, cuda, N=N, H=H, W=W, n_sent=n_sent, vocab=512, context=16)[0]
        if n_sent > 1:
            r.sentences[1].other_noun_rows = []
            r.sentences[1].n_nouns = 0
        refs.append(r)
    for clamp in ("per_ref", "persistent"):
        a, b = HybridGLPipeline(model, res=64, k_clamp=clamp), HybridGLPipeline(model, res=64, k_clamp=clamp)
        a.group_tail, b.group_tail = True, False
        for p in (a, b):
            assert p.run(iter(refs), group=19, collect=True) == len(refs)
        torch.cuda.synchronize()
        assert np.array_equal(a.partial_rows(), b.partial_rows()) and a.partial_rows().shape[0] == sum(s[3] for s in shapes) * 3 + sum(s[3] for s in shapes[:4])
        assert np.array_equal(a.winning_indices(), b.winning_indices())
        assert torch.equal(a.cum, b.cum) and int(a.cum[1]) > 0
        for (h0, t0, o0), (h1, t1, o1) in zip(a.collected, b.collected):
            for u, v in zip(o0, o1):
                assert torch.equal(torch.nan_to_num(u), torch.nan_to_num(v)) and torch.equal(torch.isnan(u), torch.isnan(v))
