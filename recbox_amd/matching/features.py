"""Matching-side feature schema (drop-in for ``recbox.matching.features.FeatureMap``).

Mirrors /root/reference/recbox/matching/features.py:12-58: ``feature_specs`` is
an ordered feature -> spec dict with ``source`` (user|item), ``type``
(numeric|categorical|sequence), ``vocab_size``, ``padding_idx`` (the matching
tokenizer pads with ``vocab_size - 1``, matching/preprocess.py:45-59),
``max_len``, ``share_embedding``, ``embedding_dim``, ``embedding_callback``.
The offline ``FeatureEncoder`` (CSV -> ids) of the same reference file is out of
scope.
"""
import json
import logging
import os
from collections import OrderedDict


class FeatureMap(object):
    def __init__(self, dataset_id, data_dir, query_index=None, corpus_index=None, label_name=None,
                 version="pytorch"):
        self.data_dir = data_dir
        self.dataset_id = dataset_id
        self.version = version
        self.num_fields = 0
        self.num_features = 0
        self.num_items = 0
        self.query_index = query_index
        self.corpus_index = corpus_index
        self.label_name = label_name
        self.feature_specs = OrderedDict()

    def load(self, json_file):
        logging.info("Load feature_map from json: " + json_file)
        with open(json_file, "r", encoding="utf-8") as fd:
            blob = json.load(fd, object_pairs_hook=OrderedDict)
        if blob["dataset_id"] != self.dataset_id:
            raise RuntimeError("dataset_id={} does not match to feature_map!".format(self.dataset_id))
        self.num_fields = blob["num_fields"]
        self.num_features = blob.get("num_features", None)
        self.label_name = blob.get("label_name", None)
        self.feature_specs = OrderedDict(blob["feature_specs"])

    def save(self, json_file):
        logging.info("Save feature_map to json: " + json_file)
        os.makedirs(os.path.dirname(json_file), exist_ok=True)
        blob = OrderedDict()
        for key in ("dataset_id", "num_fields", "num_features", "num_items", "query_index", "corpus_index",
                    "label_name", "feature_specs"):
            blob[key] = getattr(self, key)
        with open(json_file, "w", encoding="utf-8") as fd:
            json.dump(blob, fd, indent=4)

    def get_num_fields(self, feature_source=[]):
        sources = feature_source if isinstance(feature_source, list) else [feature_source]
        return sum(1 for spec in self.feature_specs.values() if not sources or spec["source"] in sources)
