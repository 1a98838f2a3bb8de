"""TransH (/root/reference/TransH.py): entities projected on the relation's hyperplane."""
from . import _lib
from .Model import Model


class TransH(Model):
    model_id = _lib.TRANSH
    table_names = ("ent_embeddings", "rel_embeddings", "normal_vectors")  # TransH.py:26-28

    def table_shapes(self):
        c = self.config
        return {"ent_embeddings": (c.entTotal, c.hidden_size), "rel_embeddings": (c.relTotal, c.hidden_size),
                "normal_vectors": (c.relTotal, c.hidden_size)}
