"""TransD (/root/reference/TransD.py): e + (e . e_p) r_p; all four tables use hidden_size (TransD.py:37-40)."""
from . import _lib
from .Model import Model


class TransD(Model):
    model_id = _lib.TRANSD
    # engine order: ent, rel, relation-side aux, entity-side aux
    table_names = ("ent_embeddings", "rel_embeddings", "rel_transfer", "ent_transfer")  # TransD.py:37-40

    def table_shapes(self):
        c = self.config
        return {"ent_embeddings": (c.entTotal, c.hidden_size), "rel_embeddings": (c.relTotal, c.hidden_size),
                "rel_transfer": (c.relTotal, c.hidden_size), "ent_transfer": (c.entTotal, c.hidden_size)}
