"""Row f-1: sampler partition (CPU) and the device-side input pipeline (GPU) against the reference's transform
(ToTensor + Normalize, data_loader_v2.py:82-86) written out in numpy."""
import numpy as np
import pytest
import torch


def test_sharded_sampler_partitions_every_epoch():
    from lct_vqa_amd.data import ShardedSampler
    n, world = 1003, 4
    for epoch in (0, 1):
        parts = []
        for r in range(world):
            s = ShardedSampler(n, r, world, seed=10)
            s.set_epoch(epoch)
            parts.append(s.indices())
        assert all(len(p) == len(parts[0]) == 251 for p in parts)
        allidx = np.concatenate(parts)
        assert set(allidx.tolist()) == set(range(n))                 # every sample seen; padding wraps around
    a, b = ShardedSampler(n, 0, 1, seed=10), ShardedSampler(n, 0, 1, seed=10)
    b.set_epoch(1)
    assert not np.array_equal(a.indices(), b.indices())              # reshuffled per epoch
    assert np.array_equal(ShardedSampler(n, 0, 1, seed=10, shuffle=False).indices(), np.arange(n))


@pytest.mark.gpu
def test_device_batcher_matches_reference_transform(hip_lib, tmp_path):
    from lct_vqa_amd.data import DeviceBatcher, MEAN, STD
    rs = np.random.RandomState(0)
    nimg, nq, S, T, A = 7, 23, 64, 14, 40
    np.save(tmp_path / 'images.npy', rs.randint(0, 256, (nimg, S, S, 3)).astype(np.uint8))
    images = np.load(tmp_path / 'images.npy', mmap_mode='r')                          # memmap stands in for HDF5
    coco_ids = np.arange(100, 100 + nimg)
    enc_qst = rs.randint(0, 50, (nq, T)).astype(np.int32)
    qst_len = rs.randint(3, T, (nq,))
    enc_ans = (rs.rand(nq, A) > 0.9).astype(np.int64)
    enc_ans[0] = 0                                                                     # no valid answer -> <unk>
    img_id = coco_ids[rs.randint(0, nimg, (nq,))]
    db = DeviceBatcher(images, coco_ids, enc_qst, qst_len, enc_ans, img_id, 8, 'cuda', unk_idx=1, shuffle=True)
    from lct_vqa_amd.basic_vqa import misc
    from lct_vqa_amd import ops
    seen = []
    hits = total = 0
    for batch in db:
        # the multiple-choice table: every valid answer id of the sample from the front, -1 padded, 10 rows
        # (data_loader_v2.py:63-67), and it feeds the accuracy counters of the train / val loops directly
        ch = batch['answer_multi_choice']
        assert ch.shape == (10, len(batch['image_id'])) and ch.dtype == torch.int64
        ans = batch['enc_ans'].cpu().numpy()
        for k in range(ch.shape[1]):
            valid = ans[k].nonzero()[0]
            exp = np.full(10, -1)
            exp[:min(10, valid.size)] = valid[:10]
            assert np.array_equal(ch[:, k].cpu().numpy(), exp)
        pred = batch['answer_label'].clone()
        pred[::2] = A + 5                                                       # half of the predictions are wrong on purpose
        ref_hits = sum(int(pred[k].item() in set(ans[k].nonzero()[0][:10].tolist())) for k in range(len(pred)))
        assert misc.num_correct(pred, ch) == ref_hits                            # reference formula misc.py:42-45
        assert misc.num_correct(pred, list(ch)) == ref_hits                      # ... and in the collate's list-of-10 form
        counts = torch.zeros(2, dtype=torch.int64, device='cuda')
        logits = torch.full((len(pred), A + 6), -1.0, device='cuda')
        logits[torch.arange(len(pred)), pred] = 1.0
        ops.eval_multi_choice(logits, ch, 1, counts)
        assert counts[0].item() == ref_hits
        hits += ref_hits
        total += len(pred)
        assert batch['image_name'][0] == 'COCO_train2014_%012d' % int(batch['image_id'][0]) and len(batch['image_name']) == len(pred)
        iid = batch['image_id'].cpu().numpy()
        for k in range(len(iid)):
            raw = np.asarray(images[int(iid[k]) - 100]).astype(np.float32) / 255.0     # ToTensor
            ref = (raw - np.array(MEAN, np.float32)) / np.array(STD, np.float32)       # Normalize
            got = batch['image'][k].cpu().numpy().transpose(1, 2, 0)
            assert np.abs(got - ref).max() < 1e-5
        assert batch['image'].dtype == torch.float32 and batch['question'].dtype == torch.int64
        seen.append(batch['question'].cpu().numpy())
        lab = batch['answer_label'].cpu().numpy()
        assert ((lab >= 0) & (lab < A)).all()
    assert 0 < hits < total
    seen = np.concatenate(seen)
    assert len(seen) == nq and sorted(map(tuple, seen.tolist())) == sorted(map(tuple, enc_qst.astype(np.int64).tolist()))
