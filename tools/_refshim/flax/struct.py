"""`flax.struct` stand-in (dev-only): frozen dataclasses with .replace and
pytree metadata (pytree_node=False fields are static)."""
import dataclasses


def field(pytree_node=True, **kwargs):
  md = dict(kwargs.pop("metadata", {}) or {})
  md["pytree_node"] = pytree_node
  return dataclasses.field(metadata=md, **kwargs)


def dataclass(cls):
  cls = dataclasses.dataclass(frozen=True)(cls)
  dyn, static = [], []
  for f in dataclasses.fields(cls):
    (dyn if f.metadata.get("pytree_node", True) else static).append(f.name)
  cls._shim_struct_fields = (dyn, static)

  def replace(self, **updates):
    return dataclasses.replace(self, **updates)

  cls.replace = replace
  return cls
