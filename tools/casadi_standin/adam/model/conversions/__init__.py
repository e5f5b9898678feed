def to_idyntree_model(*a, **k): raise NotImplementedError
