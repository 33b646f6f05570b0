"""Parameter name -> shape tables of the reference models (TEST INFRASTRUCTURE, see oracle/__init__.py).

Order and names are those of ``named_parameters()`` of the reference classes
(model/model.py:385-394,460-484; model/itm.py:12-21; probe in SURVEY.md Appendix B):
tied tensors (cls.decoder.weight = word_embeddings.weight, feat_regress.weight =
img_linear.weight) appear once."""
from collections import OrderedDict


def roberta_shapes(cfg, img_dim=2048):
    H, I, V = cfg.hidden_size, cfg.intermediate_size, cfg.vocab_size
    s = OrderedDict()
    e = "roberta.embeddings."
    s[e + "word_embeddings.weight"] = (V, H)
    s[e + "position_embeddings.weight"] = (cfg.max_position_embeddings, H)
    s[e + "new_token_type_embeddings.weight"] = (cfg.type_vocab_size, H)
    s[e + "LayerNorm.weight"] = (H,)
    s[e + "LayerNorm.bias"] = (H,)
    i = "roberta.img_embeddings."
    s[i + "img_linear.weight"] = (H, img_dim)
    s[i + "img_linear.bias"] = (H,)
    s[i + "img_layer_norm.weight"] = (H,)
    s[i + "img_layer_norm.bias"] = (H,)
    s[i + "pos_layer_norm.weight"] = (H,)
    s[i + "pos_layer_norm.bias"] = (H,)
    s[i + "pos_linear.weight"] = (H, 7)
    s[i + "pos_linear.bias"] = (H,)
    s[i + "mask_embedding.weight"] = (2, img_dim)
    s[i + "LayerNorm.weight"] = (H,)
    s[i + "LayerNorm.bias"] = (H,)
    for l in range(cfg.num_hidden_layers):
        p = "roberta.encoder.layer.%d." % l
        for n in ("query", "key", "value"):
            s[p + "attention.self.%s.weight" % n] = (H, H)
            s[p + "attention.self.%s.bias" % n] = (H,)
        s[p + "attention.output.dense.weight"] = (H, H)
        s[p + "attention.output.dense.bias"] = (H,)
        s[p + "attention.output.LayerNorm.weight"] = (H,)
        s[p + "attention.output.LayerNorm.bias"] = (H,)
        s[p + "intermediate.dense.weight"] = (I, H)
        s[p + "intermediate.dense.bias"] = (I,)
        s[p + "output.dense.weight"] = (H, I)
        s[p + "output.dense.bias"] = (H,)
        s[p + "output.LayerNorm.weight"] = (H,)
        s[p + "output.LayerNorm.bias"] = (H,)
    s["roberta.pooler.dense.weight"] = (H, H)
    s["roberta.pooler.dense.bias"] = (H,)
    return s


def pretrain_shapes(cfg, img_dim=2048, img_label_dim=1601, n_valid_ids=45):
    H, V = cfg.hidden_size, cfg.vocab_size
    s = roberta_shapes(cfg, img_dim)
    s["cls.bias"] = (V,)
    s["cls.dense.weight"] = (H, H)
    s["cls.dense.bias"] = (H,)
    s["cls.layer_norm.weight"] = (H,)
    s["cls.layer_norm.bias"] = (H,)
    s["vis_cls.bias"] = (n_valid_ids,)
    s["vis_cls.dense.weight"] = (H, H)
    s["vis_cls.dense.bias"] = (H,)
    s["vis_cls.layer_norm.weight"] = (H,)
    s["vis_cls.layer_norm.bias"] = (H,)
    s["vis_cls.decoder.weight"] = (n_valid_ids, H)
    s["feat_regress.bias"] = (img_dim,)
    s["feat_regress.net.0.weight"] = (H, H)
    s["feat_regress.net.0.bias"] = (H,)
    s["feat_regress.net.2.weight"] = (H,)
    s["feat_regress.net.2.bias"] = (H,)
    s["region_classifier.net.0.weight"] = (H, H)
    s["region_classifier.net.0.bias"] = (H,)
    s["region_classifier.net.2.weight"] = (H,)
    s["region_classifier.net.2.bias"] = (H,)
    s["region_classifier.net.3.weight"] = (img_label_dim, H)
    s["region_classifier.net.3.bias"] = (img_label_dim,)
    s["itm_output.weight"] = (2, H)
    s["itm_output.bias"] = (2,)
    return s


def itm_rank_shapes(cfg, img_dim=2048):
    H = cfg.hidden_size
    s = roberta_shapes(cfg, img_dim)
    s["itm_output.weight"] = (2, H)
    s["itm_output.bias"] = (2,)
    s["rank_output.weight"] = (1, H)
    s["rank_output.bias"] = (1,)
    return s
