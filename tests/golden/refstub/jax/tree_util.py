def register_pytree_node_class(cls):
    return cls
