"""TransR (/root/reference/TransR.py): per-relation [ent_size x rel_size] projection matrices."""
from . import _lib
from .Model import Model


class TransR(Model):
    model_id = _lib.TRANSR
    table_names = ("ent_embeddings", "rel_embeddings", "transfer_matrix")  # TransR.py:29-31

    def dims(self):
        c = self.config
        return c.ent_size, c.rel_size

    def table_shapes(self):
        c = self.config
        return {"ent_embeddings": (c.entTotal, c.ent_size), "rel_embeddings": (c.relTotal, c.rel_size),
                "transfer_matrix": (c.relTotal, c.ent_size * c.rel_size)}
