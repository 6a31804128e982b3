"""Byte ledger of the headline training step (Tmall shape, 4096 sessions, fp32 parity mode or bf16 mode): every HBM tensor of the step with its
producer, its consumers and its bytes, from the shapes alone -- to be held against the PMC total of the same step
(profiles/r04_pmc_traffic*.json).  "needed" = the kernel decomposition of this build cannot avoid the transfer (a tensor that one
kernel writes and a LATER kernel reads crosses HBM: the step's working set is 3 GB, the L2s hold 32 MB, the Infinity Cache 256 MB);
"avoidable" = a different fusion inside the current decomposition would remove it.
usage: python tools/byte_ledger.py [f32|bf16] [pmc.json]      (no GPU needed)"""
import json
import sys

mode = sys.argv[1] if len(sys.argv) > 1 else 'f32'
pmc = sys.argv[2] if len(sys.argv) > 2 else None
bf = mode == 'bf16'
B, L, H, I, K = 4096, 50, 20, 30, 3
d_id = d_im = d_u = d_s = d_c = d_int = 64
d_i = d_id + d_im
M = B * L
rows_enc = int(B * (1 + H) / 2)            # packed history rows per encoder: history_len ~ U{1..20}
dm = 128
items = 1000000
MB = 1e6
rowsT = []                                  # (group, tensor, MB per access, writes, reads, producer, consumers, verdict)


def t(group, name, elems, w, r, prod, cons, verdict, esz=4):
    rowsT.append((group, name, elems * esz / MB, w, r, prod, cons, verdict))


def tower(tag, d, fused_fwd):
    u = M * d
    h = 2 if bf else 4                                   # bf16 mode stores the matmul-only activations as bf16
    g = 'tower %s (d = %d)' % (tag, d)
    t(g, 'X0 (tower input rows)', u, 1, 3 if not fused_fwd else 2, 'gather_rows x2' if tag == 'item' else 'linear_smallk', 'q/k/v GEMM, W2 residual, q/k/v wgrad', 'needed')
    t(g, 'QKV stash', 3 * u, 1, 2, 'q/k/v GEMM' if not fused_fwd else 'tower_fwd_fused', 'attention fwd (not when fused), attention bwd', 'needed (bwd); the fwd read is avoided by the one-kernel layer', h if fused_fwd or bf else 4)
    if fused_fwd:
        rowsT[-1] = rowsT[-1][:4] + (1,) + rowsT[-1][5:]
    t(g, 'A (attention output)', u, 1, 2 if not fused_fwd else 1, 'attention fwd', 'W1 GEMM (not when fused), W1 wgrad', 'needed (wgrad)', h if fused_fwd or bf else 4)
    t(g, 'R1 (relu output)', u, 1, 3 if not fused_fwd else 2, 'W1 GEMM', 'W2 GEMM (not when fused), relu mask of dF1, W2 wgrad', 'mask read avoidable (sign bits: -1 read of u/32)', h if fused_fwd or bf else 4)
    t(g, 'x-hat + rstd (LayerNorm stash; the layer output itself is NOT stored)', u + M, 1, 2, 'W2 GEMM epilogue', 'pooling fwd, pooling + LayerNorm bwd', 'needed')
    t(g, 'dZ (gradient behind the LayerNorm)', u, 1, 3, 'xatt_pool_ln_bwd', 'W2 wgrad, dF1 GEMM, residual add of the dX GEMM', 'one read avoidable (one-pass dgrad + wgrad)')
    t(g, 'dF1', u, 1, 2, 'dF1 GEMM', 'W1 wgrad, dA GEMM', 'one read avoidable (one-pass)', h if bf else 4)
    t(g, 'dA (dO of the attention)', u, 1, 1, 'dA GEMM', 'attention bwd', 'avoidable (fuse the W1 data gradient into the attention backward)', h if bf else 4)
    t(g, 'dQKV', 3 * u, 1, 2, 'attention bwd', 'q/k/v wgrad, dX GEMM (K = 3d)', 'one read avoidable (one-pass)', h if bf else 4)
    t(g, 'dX0', u, 1, 1, 'dX GEMM', 'embedding scatter' if tag == 'item' else 'score-embedding wgrad', 'needed')
    nslab = 256
    t(g, 'weight-gradient slabs (5 weights x 256 partials)', 5 * nslab * d * d, 1, 1, 'wgrad kernels', 'slab_reduce_batch', 'avoidable in part (fewer partials / in-kernel last-workgroup reduction)')


tower('item', d_i, False)
tower('score', d_s, not bf or True)
for e, name in ((0, 'session-history encoder'), (1, 'item-history encoder')):
    u = rows_enc * dm
    g = '%s (BERT4Rec, %d packed rows x 128)' % (name, rows_enc)
    t(g, 'E0 (input rows: table gather + intent linear + position)', u, 1, 3, 'gather / linear kernels', 'block 0 fwd, block 0 bwd (q/k/v wgrad operand), dX chain', 'needed')
    t(g, 'block 0 stash: QKV, x-hat1, F1 (relu), x-hat2 (+ row stats)', 6 * u, 1, 1, 'enc_block_fwd', 'enc_block_bwd', 'F1 / x-hat avoidable by recompute (-2 of 6)')
    t(g, 'block 0 output + LayerNorm-1 output C', 2 * u, 1, 2, 'enc_block_fwd', 'enc_block_bwd, wgrad operands', 'needed')
    t(g, "[K'|V'] of the pruned last block", 2 * u, 1, 2, 'enc_block_fwd', 'enc_last_fwd, enc_last_bwd', 'needed')
    t(g, 'backward rows for the wgrad kernels: dZ2, dF1, dZ1, dQKV (3), dKV (2)', 8 * u, 1, 1.25, 'enc_block_bwd / enc_last_bwd', 'wgrad_b3 (6 products), gemm_rows_b3k (dX, K = 384 / 256)', 'needed by the split data / weight gradient kernels')
    t(g, 'dE0 (+ dX partials of the two K > 128 products)', 2 * u, 1, 1, 'gemm_rows_b3k', 'scatter, intent wgrad', 'needed')
    t(g, 'weight-gradient slabs (12 weights x <= 256 partials)', 12 * 256 * dm * dm // 2, 1, 1, 'wgrad kernels', 'slab_reduce_batch', 'avoidable in part')
g = 'embedding tables'
t(g, 'iid_embeddings rows gathered (lists + item histories, uniform ids: no reuse)', (M + rows_enc) * d_id, 0, 1, 'Adam', 'gather_rows', 'needed (the algorithmic gather)')
t(g, 'iid gradient rows (atomic read-modify-write) + row flags', (M + rows_enc) * d_id, 1, 1, 'scatter_add_rows', 'adam_rows', 'needed')
t(g, 'iid table dense Adam: p, m, v read + written, g read / cleared in flagged rows only', items * d_id * 6 + (M + rows_enc) * d_id * 2, 1, 0, 'adam_rows', 'next step', 'needed by the reference\'s dense semantics (lazy form: bit-identical, touched rows only -- loses at 29 % touched rows)')
t(g, 'user / context / class tables, their gradients and Adam (100 k x 64, 931 x 64, 357 x 64)', (100000 + 931 + 357) * 64 * 8, 1, 0, 'adam', '', 'needed')
g = 'session head, loss, inputs'
t(g, 'batch inputs: ids, scores (fp32 + fp64 copy), his_intents, labels', B * (L * (2 * 4 + K * 12 + 4) + H * (I * 4 + 8) + 64), 0, 1, 'feed', 'forward, losses', 'needed (algorithmic)', 1)
t(g, 'outputs weights / ens_score / intents + their gradients', B * (L * (K + 1) + I) * 2, 1, 1, 'ens kernels, loss kernels', 'loss kernels, ens_bwd', 'needed')
t(g, 'B-row tensors of the head (PREDIN, FEAT, QV, QK, XBAR, their gradients; ~40 tensors of B x <= 384)', B * 384 * 40 // 3, 1, 1.5, 'B-row GEMMs', 'B-row GEMMs / wgrads', 'needed (small)')
t(g, 'packed weight images (fp32 fragments + bf16 three-plane images), once per step', 2.2e6, 1, 1, 'pack kernels', 'every GEMM (L2-resident afterwards)', 'needed')

tot = 0.0
by_verdict = {}
print('| group | tensor | MB per pass | writes | reads | MB per step | producer -> consumers | verdict |')
print('|---|---|---|---|---|---|---|---|')
for (g, name, mb, w, r, prod, cons, verdict) in rowsT:
    step = mb * (w + r)
    tot += step
    print('| %s | %s | %.1f | %g | %g | %.0f | %s -> %s | %s |' % (g, name, mb, w, r, step, prod, cons, verdict))
print()
print('analytic total: %.2f GB per step (%s mode)' % (tot / 1e3, mode))
if pmc:
    j = json.load(open(pmc))
    print('PMC total (%s): %.2f GB per step; analytic / PMC = %.2f' % (pmc, j['hbm_GB_per_train_step'], tot / 1e3 / j['hbm_GB_per_train_step']))
