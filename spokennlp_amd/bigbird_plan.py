"""Block lists of BigBird block-sparse attention (host side, integers only).

[hf] models/big_bird/modeling_big_bird.py `BigBirdBlockSparseAttention.bigbird_block_sparse_attention` computes five
query-block groups against concatenations of key blocks.  Restated as "query block i visits this list of key blocks":

  i = 0, nb-1          every key block                                   ("1st PART" / "5th PART": global rows)
  i = 1                [0, 1, 2, nb-1] + rand[h][0]                      ("2nd PART")
  2 <= i <= nb-3       [0] + [i-1, i, i+1] + rand[h][i-1] + [nb-1]       ("3rd PART": global, sliding, random, global)
  i = nb-2             [0, nb-3, nb-2, nb-1] + rand[h][nb-3]             ("4th PART")

ONE softmax runs over the concatenation, so a key block that appears twice counts twice (in eval mode the reference's
random blocks are all block 0: `_bigbird_block_rand_mask*` return zeros when `not self.training`, i.e. block 0 is counted
1 + num_random_blocks times).  The lists keep those multiplicities; libamdseg's amdseg_attn_list_* stream them through an
online softmax, which is the same sum.

The random plan itself (`rand[h]`, an int array [nb-2, r] per head) is the reference dependency's own numpy-seeded
procedure: `np.random.seed(layer_idx)` followed by `_bigbird_block_rand_mask` (L in {1024, 3072, 4096}) or
`_get_rand_attn_plan` + `_bigbird_block_rand_mask_with_head`.  A fine-tuned checkpoint is only meaningful with exactly that
plan, so `rand_blocks` calls those (pure, static-like) functions of the installed `transformers` -- the same package the
reference imports BigBirdModel from -- instead of re-deriving numpy's permutation stream.
"""
import numpy as np
from transformers.models.big_bird.modeling_big_bird import BigBirdBlockSparseAttention as _Ref    # at import time: importing
# transformers sub-modules lazily inside a training step would consume the python `random` stream the CSSL sampling uses

BLOCK = 64


def rand_blocks(seq_len, num_heads, num_rand_blocks, seed, training, max_seqlen, block=BLOCK):
    """[num_heads, seq_len/block - 2, num_rand_blocks] int32: the reference's random key blocks for layer `seed`."""
    nb = seq_len // block
    if not training:                                   # reference: "During inference (eval) no randomness" -> zeros
        return np.zeros((num_heads, nb - 2, num_rand_blocks), dtype=np.int32)
    Ref = _Ref

    class _Self:                                       # the only state the plan functions read
        pass
    me = _Self()
    me.training = True
    me.max_seqlen = max_seqlen
    me._get_single_block_row_attention = Ref._get_single_block_row_attention
    np_state = np.random.get_state()
    try:
        np.random.seed(seed)
        if seq_len in (1024, 3072, 4096):
            ra = [Ref._bigbird_block_rand_mask(me, max_seqlen, max_seqlen, block, block, num_rand_blocks, last_idx=1024)[: nb - 2]
                  for _ in range(num_heads)]
        else:
            plan_len, plan_r = Ref._get_rand_attn_plan(seq_len, block, num_rand_blocks)
            ra = Ref._bigbird_block_rand_mask_with_head(me, from_seq_length=seq_len, to_seq_length=seq_len, from_block_size=block,
                                                        to_block_size=block, num_heads=num_heads, plan_from_length=plan_len,
                                                        plan_num_rand_blocks=plan_r)
    finally:
        np.random.set_state(np_state)                  # the reference leaves the global stream reseeded; a library should not
    return np.stack(ra, axis=0).astype(np.int32)


def key_lists(nb, rand):
    """rand: [heads, nb-2, r].  Returns (klist [heads, nb, nb] int32 padded with 0, kcnt [heads, nb] int32)."""
    heads, _, r = rand.shape
    if nb < 5:
        raise ValueError("block-sparse attention needs at least 5 blocks")
    klist = np.zeros((heads, nb, nb), dtype=np.int32)
    kcnt = np.zeros((heads, nb), dtype=np.int32)
    for h in range(heads):
        for i in range(nb):
            if i == 0 or i == nb - 1:
                row = list(range(nb))
            elif i == 1:
                row = [0, 1, 2, nb - 1] + [int(x) for x in rand[h, 0]]
            elif i == nb - 2:
                row = [0, nb - 3, nb - 2, nb - 1] + [int(x) for x in rand[h, nb - 3]]
            else:
                row = [0, i - 1, i, i + 1] + [int(x) for x in rand[h, i - 1]] + [nb - 1]
            if len(row) > nb:
                raise ValueError("block list longer than the number of blocks")
            klist[h, i, :len(row)] = row
            kcnt[h, i] = len(row)
    return klist, kcnt


def transpose_lists(klist, kcnt):
    """Per (head, key block): the query blocks that visit it, with the same multiplicities.  Row length may exceed nb
    (block 0 is visited by every query block, several times in eval mode), so the stride is returned."""
    heads, nb, _ = klist.shape
    rows = [[[] for _ in range(nb)] for _ in range(heads)]
    for h in range(heads):
        for i in range(nb):
            for t in range(int(kcnt[h, i])):
                rows[h][int(klist[h, i, t])].append(i)
    stride = max(len(r) for hr in rows for r in hr)
    qlist = np.zeros((heads, nb, stride), dtype=np.int32)
    qcnt = np.zeros((heads, nb), dtype=np.int32)
    for h in range(heads):
        for k in range(nb):
            qlist[h, k, :len(rows[h][k])] = rows[h][k]
            qcnt[h, k] = len(rows[h][k])
    return qlist, qcnt, stride


def build(seq_len, num_heads, num_rand_blocks, seed, training, max_seqlen):
    """All four tables with ONE common row stride (the C ABI takes a single list_stride)."""
    nb = seq_len // BLOCK
    rand = rand_blocks(seq_len, num_heads, num_rand_blocks, seed, training, max_seqlen)
    klist, kcnt = key_lists(nb, rand)
    qlist, qcnt, stride = transpose_lists(klist, kcnt)
    stride = max(stride, nb)
    kl = np.zeros((num_heads, nb, stride), dtype=np.int32)
    kl[:, :, :nb] = klist
    ql = np.zeros((num_heads, nb, stride), dtype=np.int32)
    ql[:, :, :qlist.shape[2]] = qlist
    # launch order: block indices by decreasing list length (stable), per head
    korder = np.argsort(-kcnt, axis=1, kind="stable").astype(np.int32)
    qorder = np.argsort(-qcnt, axis=1, kind="stable").astype(np.int32)
    return dict(klist=kl, kcnt=kcnt, qlist=ql, qcnt=qcnt, stride=stride, rand=rand, korder=korder, qorder=qorder)


def min_block_sparse_len(num_rand_blocks, block=BLOCK):
    """Sequences up to this length run full attention in the reference (BigBirdModel.forward: max_tokens_to_attend)."""
    return (5 + 2 * num_rand_blocks) * block
