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


# scalar header fields of feature_map.json, in file order, with the value a fresh map starts from
_HEADER = (("num_fields", 0), ("total_features", 0), ("input_length", 0), ("labels", None), ("group_id", None))


class FeatureMap(object):
    def __init__(self, dataset_id, data_dir):
        self.dataset_id, self.data_dir = dataset_id, data_dir       # data_dir: where pretrained tables are looked up
        for key, start in _HEADER:
            setattr(self, key, [] if key == "labels" else start)
        self.features = OrderedDict()
        self.column_index = dict()
        self.default_emb_dim = None

    # ---- persistence (the reference's feature_map.json layout: header scalars + a list of one-entry dicts) ----
    def load(self, json_file, params):
        logging.info("Load feature_map from json: " + json_file)
        with io.open(json_file, "r", encoding="utf-8") as fd:
            doc = json.load(fd)
        if doc["dataset_id"] != self.dataset_id:
            raise RuntimeError("dataset_id={} does not match feature_map!".format(self.dataset_id))
        self.num_fields = doc["num_fields"]                          # the only header field that must be present
        for key, start in _HEADER[1:]:
            setattr(self, key, doc.get(key, [] if key == "labels" else start))
        self.default_emb_dim = params.get("embedding_dim", None)
        specs = OrderedDict()
        for entry in doc["features"]:
            specs.update(entry)
        keep = params.get("use_features", None)
        self.features = OrderedDict((name, specs[name]) for name in keep) if keep else specs
        overrides = params.get("feature_specs", None)
        if overrides:
            self.update_feature_specs(overrides)
        self.set_column_index()

    def update_feature_specs(self, feature_specs):
        for patch in feature_specs:
            settings = dict((k, v) for k, v in patch.items() if k != "name")
            for name in _as_list(patch["name"]):
                self.features[name].update(settings)

    def save(self, json_file):
        logging.info("Save feature_map to json: " + json_file)
        os.makedirs(os.path.dirname(json_file), exist_ok=True)
        doc = OrderedDict(dataset_id=self.dataset_id)
        for key, _ in _HEADER:
            if key == "group_id" and self.group_id is None:
                continue
            doc[key] = getattr(self, key)
        doc["features"] = [{name: spec} for name, spec in self.features.items()]
        with open(json_file, "w") as fd:
            json.dump(doc, fd, indent=4)

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
        """feature -> column (or list of ``max_len`` columns) of the flat [B, cols] batch; labels follow the inputs."""
        logging.info("Set column index...")
        cursor = 0
        for name, spec in self.features.items():
            width = spec.get("max_len")
            self.column_index[name] = cursor if width is None else list(range(cursor, cursor + width))
            cursor += 1 if width is None else width
        self.input_length = cursor
        self.column_index.update((label, cursor + k) for k, label in enumerate(self.labels))

    def get_column_index(self, feature):
        if feature not in self.column_index:
            self.set_column_index()
        return self.column_index[feature]
