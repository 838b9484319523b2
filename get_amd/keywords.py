"""The ``**kargs`` vocabulary of the model boundary (string values of the reference's
``setting_keywords.KeyWordSettings``).  Only the keys the hot path or its
caller protocol touches are listed; the strings are the contract, the attribute names mirror
upstream so call sites read the same."""


class KeyWordSettings(object):
    # consumed by Graph_basedSemantiStructure.forward
    Query_lens = "query_lens"                       # (B,) number of unique claim nodes
    Doc_lens = "docs_lens"                          # presence only
    DocLensIndices = "doc_lens_indices"             # 3-tuple, [2] used for a shape assert
    QueryLensIndices = "query_lens_indices"
    DocContentNoPaddingEvidence = "doc_content_without_padding_evidences"   # (B1,R) node ids
    QueryContentNoPaddingEvidence = "query_content_without_padding_evidences"
    Evd_Docs_Adj = "docs_adj"                       # (B1,R,R) dense or PackedAdj
    Query_Adj = "query_adj"                         # (B,L,L) dense or PackedAdj
    EvidenceCountPerQuery = "evd_cnt_each_query"    # (B,) int
    FIXED_NUM_EVIDENCES = "fixed_num_evidences"     # int (30)
    DocSources = "doc_sources"                      # (B,n) int, -1 = padding
    QuerySources = "query_sources"                  # (B,1) int
    OutputRankingKey = "output_ranking"
    GNN_Window = "gnn_window"
    # passed through by the fitter and ignored by forward
    TempLabel = "fc_labels"
    UseCuda = "use_cuda"

    class FCClass:
        CharSourceKey = "char_source"
        QueryCharSource = "query_char_source"
        DocCharSource = "doc_char_source"
        DocAttentionScore = "doc_attention_score"
        WordAttentionScore = "word_attention_score"
