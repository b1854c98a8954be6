"""Shared test helpers: seeded input generators that mirror tools/make_golden.py."""
import math

import torch

ERB_KEYS = (
    'rbr_3x3_branch.weight', 'rbr_3x3_branch.bias', 'rbr_3x1_branch.weight', 'rbr_3x1_branch.bias',
    'rbr_1x3_branch.weight', 'rbr_1x3_branch.bias', 'rbr_1x1_3x3_1x1_branch_1x1_1.weight',
    'rbr_1x1_3x3_1x1_branch_3x3.weight', 'rbr_1x1_3x3_1x1_branch_1x1_2.weight')


def _rand(gen, *shape, scale=1.0):
    return (torch.rand(*shape, generator=gen) * 2 - 1) * scale


def erb_inputs(C, O, seed):
    """Same seeded branch weights as tools/make_golden.py::erb_inputs."""
    g = torch.Generator().manual_seed(seed)
    return {
        'rbr_3x3_branch.weight': _rand(g, O, C, 3, 3, scale=1 / math.sqrt(9 * C)),
        'rbr_3x3_branch.bias': _rand(g, O, scale=1 / math.sqrt(9 * C)),
        'rbr_3x1_branch.weight': _rand(g, O, C, 3, 1, scale=1 / math.sqrt(3 * C)),
        'rbr_3x1_branch.bias': _rand(g, O, scale=1 / math.sqrt(3 * C)),
        'rbr_1x3_branch.weight': _rand(g, O, C, 1, 3, scale=1 / math.sqrt(3 * C)),
        'rbr_1x3_branch.bias': _rand(g, O, scale=1 / math.sqrt(3 * C)),
        'rbr_1x1_3x3_1x1_branch_1x1_1.weight': _rand(g, 2 * C, C, 1, 1, scale=1 / math.sqrt(C)),
        'rbr_1x1_3x3_1x1_branch_3x3.weight': _rand(g, O, 2 * C, 3, 3, scale=1 / math.sqrt(18 * C)),
        'rbr_1x1_3x3_1x1_branch_1x1_2.weight': _rand(g, O, O, 1, 1, scale=1 / math.sqrt(O)),
    }
