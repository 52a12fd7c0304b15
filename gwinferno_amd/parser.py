"""The containers ``construct_hierarchical_model`` consumes (gwinferno/pipeline/parser.py:13-31): a population
model class with the names of its hyper-parameters, and a prior class with its arguments.  The YAML reader that
fills them in the reference (parser.py:48-163) is configuration plumbing and not part of this package: build the
two dictionaries directly (see ``gwinferno_amd.likelihood.construct_hierarchical_model``)."""


class PopModel(object):
    def __init__(self, model, params):
        self.model = model
        self.params = params


class PopPrior(object):
    def __init__(self, dist, params):
        self.dist = dist
        self.params = params


class PopMixtureModel(PopModel):
    def __init__(self, model, mix_dist, mix_params, components, component_params):
        self.model = model
        self.components = components
        self.mixing_dist = mix_dist
        self.mixing_params = mix_params
        self.component_params = component_params
