"""TransE (/root/reference/TransE.py): |l2n(h) + l2n(r) - l2n(t)|_1, margin-ranking loss."""
from . import _lib
from .Model import Model


class TransE(Model):
    model_id = _lib.TRANSE
    table_names = ("ent_embeddings", "rel_embeddings")  # TransE.py:21-22

    def table_shapes(self):
        c = self.config
        return {"ent_embeddings": (c.entTotal, c.hidden_size), "rel_embeddings": (c.relTotal, c.hidden_size)}
