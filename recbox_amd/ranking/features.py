"""Ranking-side feature schema (drop-in for ``recbox.ranking.features.FeatureMap``).

Mirrors /root/reference/recbox/ranking/features.py:25-125: an ordered
feature -> spec dict (``type`` numeric|categorical|sequence|meta, ``source``,
``vocab_size``, ``padding_idx``, ``max_len``, ``share_embedding``,
``embedding_dim``, ``feature_encoder``, ``pretrained_emb`` ...), JSON load/save
in the reference's on-disk format, and the feature -> column(s) map of the flat
``[B, cols]`` batch tensor (sequence features take ``max_len`` columns).
Pure host-side schema: nothing here touches the GPU.
"""
import io
import json
import logging
import os
from collections import OrderedDict


def _as_list(x):
    return x if isinstance(x, list) else [x]


class FeatureMap(object):
    def __init__(self, dataset_id, data_dir):
        self.data_dir = data_dir
        self.dataset_id = dataset_id
        self.num_fields = 0
        self.total_features = 0
        self.input_length = 0
        self.features = OrderedDict()
        self.labels = []
        self.column_index = dict()
        self.group_id = None
        self.default_emb_dim = None

    # ---- persistence (same JSON layout as the reference) ----
    def load(self, json_file, params):
        logging.info("Load feature_map from json: " + json_file)
        with io.open(json_file, "r", encoding="utf-8") as fd:
            blob = json.load(fd)
        if blob["dataset_id"] != self.dataset_id:
            raise RuntimeError("dataset_id={} does not match feature_map!".format(self.dataset_id))
        self.num_fields = blob["num_fields"]
        self.labels = blob.get("labels", [])
        self.total_features = blob.get("total_features", 0)
        self.input_length = blob.get("input_length", 0)
        self.group_id = blob.get("group_id", None)
        self.default_emb_dim = params.get("embedding_dim", None)
        self.features = OrderedDict((k, v) for entry in blob["features"] for k, v in entry.items())
        if params.get("use_features", None):
            self.features = OrderedDict((name, self.features[name]) for name in params["use_features"])
        if params.get("feature_specs", None):
            self.update_feature_specs(params["feature_specs"])
        self.set_column_index()

    def update_feature_specs(self, feature_specs):
        for col in feature_specs:
            for name in _as_list(col["name"]):
                for key, value in col.items():
                    if key != "name":
                        self.features[name][key] = value

    def save(self, json_file):
        logging.info("Save feature_map to json: " + json_file)
        os.makedirs(os.path.dirname(json_file), exist_ok=True)
        blob = OrderedDict()
        blob["dataset_id"] = self.dataset_id
        blob["num_fields"] = self.num_fields
        blob["total_features"] = self.total_features
        blob["input_length"] = self.input_length
        blob["labels"] = self.labels
        if self.group_id is not None:
            blob["group_id"] = self.group_id
        blob["features"] = [{k: v} for k, v in self.features.items()]
        with open(json_file, "w") as fd:
            json.dump(blob, fd, indent=4)

    # ---- queries ----
    def _selected(self, feature_source):
        sources = _as_list(feature_source)
        for name, spec in self.features.items():
            if spec["type"] == "meta":
                continue
            if len(sources) == 0 or spec.get("source") in sources:
                yield name, spec

    def get_num_fields(self, feature_source=[]):
        return sum(1 for _ in self._selected(feature_source))

    def sum_emb_out_dim(self, feature_source=[]):
        total = 0
        for _, spec in self._selected(feature_source):
            total += spec.get("emb_output_dim", spec.get("embedding_dim", self.default_emb_dim))
        return total

    def set_column_index(self):
        logging.info("Set column index...")
        col = 0
        for name, spec in self.features.items():
            if "max_len" in spec:
                self.column_index[name] = list(range(col, col + spec["max_len"]))
                col += spec["max_len"]
            else:
                self.column_index[name] = col
                col += 1
        self.input_length = col
        for label in self.labels:
            self.column_index[label] = col
            col += 1

    def get_column_index(self, feature):
        if feature not in self.column_index:
            self.set_column_index()
        return self.column_index[feature]
