"""CPU restatement of next-token selection (the parity ORACLE for seedmi_sample_token_bf16).

TEST INFRASTRUCTURE (see oracle/seed_oracle.py's header for the import rules).

The reference samples through `model.generate(do_sample=True, top_p=0.5, temperature=1.0, ...)`
(scripts/seed_llama_inference_8B.py:81-87, 98-105): third-party code (transformers == 4.30.2, not under /root/reference):
TemperatureLogitsWarper (scores / t), TopPLogitsWarper (ascending sort, softmax, cumsum, drop cumulative <= 1 - top_p, keep at
least one) and torch.multinomial.  Restated here in float64 with the total order the HIP kernel uses (descending probability,
ties by ascending token id) and an explicit uniform u for the draw; pinned against the installed transformers' warpers in
tests/test_sampling.py.
"""
import numpy as np


def weights(logits_row: np.ndarray, temperature: float) -> np.ndarray:
    # the kernel scales the bf16 logit by 1/temperature in fp32 before subtracting the maximum
    x = (logits_row.astype(np.float32) * np.float32(1.0 / temperature)).astype(np.float64)
    return np.exp(x - x.max())


def rank_order(w: np.ndarray) -> np.ndarray:
    """Token ids in sampling order: descending weight, ties by ascending id."""
    return np.lexsort((np.arange(w.size), -w))


def top_p_keep(logits_row: np.ndarray, temperature: float, top_p: float):
    """(order, n_keep, margin): the first n_keep ids of `order` survive; margin = distance of the nucleus boundary test to
    flipping, relative to the total mass (fp32 device sums differ from these float64 sums by ~1e-6)."""
    w = weights(logits_row, temperature)
    order = rank_order(w)
    ws = w[order]
    before = np.concatenate(([0.0], np.cumsum(ws)[:-1]))
    Z = ws.sum()
    if top_p >= 1.0:
        return order, int((ws > 0).sum()), 1.0
    keep = before < top_p * Z
    keep[0] = True
    n = int(keep.sum())
    margin = np.min(np.abs(before[max(n - 1, 0):n + 1] - top_p * Z)) / Z if n < ws.size else 1.0
    return order, n, float(margin)


def sample_token(logits_row: np.ndarray, temperature: float, top_p: float, u: float):
    """(token, margin): inverse CDF over the kept tokens in rank order at u in [0,1); margin = relative distance of the draw (and
    of the nucleus boundary) from the nearest decision boundary."""
    w = weights(logits_row, temperature)
    order, n, m_keep = top_p_keep(logits_row, temperature, top_p)
    ws = w[order][:n]
    cum = np.cumsum(ws)
    r = u * cum[-1]
    j = int(np.searchsorted(cum, r, side="right"))
    j = min(j, n - 1)
    edges = np.concatenate(([0.0], cum))
    m_draw = float(np.min(np.abs(edges - r)) / w.sum())
    return int(order[j]), min(m_keep, m_draw)


def greedy_token(logits_row: np.ndarray) -> int:
    return int(np.argmax(logits_row.astype(np.float32)))          # first index on ties
