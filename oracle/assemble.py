"""TEST INFRASTRUCTURE ONLY -- host-side batch assembly used by the oracle and the fixtures.

Turns a raw synthetic batch (``get_amd.synth.make_raw_batch``) into the dense
tensors the reference model consumes, the way the reference's data layer and
fitter do (interactions.py:295-351 for the graphs;
Fitting/FittingFC/char_man_fitter_query_repr1.py:204-250 for the de-padding and
the kargs protocol).  ``convert_text_fn`` is either the oracle's or the
reference's own (golden generation).
"""
import numpy as np


def assemble_inputs(raw: dict, cfg, convert_text_fn) -> dict:
    B, n, L, R = cfg.batch, cfg.fixed_num_evidences, cfg.len_left, cfg.len_right
    counts = raw["evd_counts"]
    q_ids = np.zeros((B, L), np.int64)
    q_adj = np.zeros((B, L, L), np.float64)
    q_len = np.zeros((B,), np.int64)
    for b in range(B):
        ids, adj, k = convert_text_fn([int(t) for t in raw["claim_tokens"][b]], L, int(raw["claim_len"][b]), cfg.window)
        q_ids[b], q_adj[b], q_len[b] = np.asarray(ids), adj, k
    b1 = int(counts.sum())
    d_ids = np.zeros((b1, R), np.int64)
    d_adj = np.zeros((b1, R, R), np.float64)
    d_len = np.zeros((b1,), np.int64)
    for i in range(b1):
        ids, adj, k = convert_text_fn([int(t) for t in raw["evd_tokens"][i]], R, int(raw["evd_len"][i]), cfg.window)
        d_ids[i], d_adj[i], d_len[i] = np.asarray(ids), adj, k
    document = np.zeros((B, n, R), np.int64)
    docs_lens = np.zeros((B, n), np.int64)
    last = 0
    for b in range(B):
        c = int(counts[b])
        document[b, :c] = d_ids[last:last + c]
        docs_lens[b, :c] = d_len[last:last + c]
        last += c
    return dict(query=q_ids, query_adj=q_adj, query_lens=q_len, doc_ids=d_ids, doc_adj=d_adj, doc_lens=d_len,
                document=document, docs_lens=docs_lens, evd_counts=counts.astype(np.int64),
                doc_sources=raw["doc_sources"], query_sources=raw["query_sources"], labels=raw["labels"])


def reference_kargs(inp: dict, torch, output_ranking=False) -> dict:
    """The ``**kargs`` the fitter passes to ``net(...)`` (char_man_fitter_query_repr1.py:234-250).

    Key strings are the values of setting_keywords.KeyWordSettings.
    """
    t = torch.from_numpy
    d_len = inp["doc_lens"]
    # torch_utils.py:145-168 get_sorted_index_and_reverse_index: numpy's DEFAULT argsort (not stable; the tie order among
    # equal lengths is whatever that algorithm yields -- pinned by fixture G9, which a stable sort did not reproduce) and
    # the inverse permutation
    order = np.argsort(-d_len)
    restore = np.argsort(order)
    k = {
        "query_lens": t(inp["query_lens"]),
        "docs_lens": inp["docs_lens"],
        "doc_lens_indices": (t(order), t(restore), t(d_len)),
        "query_lens_indices": (None, None, t(inp["query_lens"])),
        "query_sources": t(inp["query_sources"]),
        "doc_sources": t(inp["doc_sources"]),
        "fc_labels": t(inp["labels"]),
        "doc_content_without_padding_evidences": t(inp["doc_ids"]),
        "evd_cnt_each_query": t(inp["evd_counts"]),
        "fixed_num_evidences": inp["document"].shape[1],
        "query_adj": t(inp["query_adj"]),
        "docs_adj": t(inp["doc_adj"]),
    }
    if output_ranking:
        k["output_ranking"] = True
    return k
