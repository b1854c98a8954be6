"""Host-side mirror of the reference's model surface for the hot path (model.py:174-188, 303-625):
`Generator(**kargs)` / `NeRVBlock(**kargs)` with the same kwarg names, attribute names, state-dict
keys and seeded initialisation order, dispatching every computation to liborn.so.

Parameters live in ordinary nn.Linear / nn.Conv2d holder modules (never called) so that
`state_dict()` keys, `hasattr(layer, 'rbr_3x3_branch')` probes (main_eval.py:248-249) and the
default initialisers match the reference one for one.
"""
import torch
import torch.nn as nn

from . import ops

_SUPPORTED_BRANCH = ('NeRV_vanilla', 'ERB')


def _only(value, allowed, what):
    if value not in allowed:
        raise NotImplementedError(f'{what}={value!r} is outside the MI355X hot path (built: {allowed}); '
                                  'see SURVEY.md section 8 for the scope')


class NeRVBlock(nn.Module):
    """model.py:303-567.  ERB: 3x3 + 3x1 + 1x3 + (1x1 -> 3x3 -> 1x1), merged online every forward."""

    def __init__(self, **kargs):
        super().__init__()
        self.ngf, self.new_ngf, self.stride = kargs['ngf'], kargs['new_ngf'], kargs['stride']
        self.deploy = kargs['deploy']
        self.branch_type = kargs['branch_type']
        _only(kargs.get('norm', 'none'), ('none',), 'norm')
        _only(kargs.get('act', 'swish'), ('swish',), 'act')
        self.out_channels = self.new_ngf * self.stride * self.stride
        C, O = self.ngf, self.out_channels
        if self.deploy:
            self.rbr_reparam = nn.Conv2d(C, O, (3, 3), 1, 1, bias=True)
        else:
            _only(self.branch_type, _SUPPORTED_BRANCH, 'branch_type')
            if self.branch_type == 'NeRV_vanilla':
                self.branch = nn.Conv2d(C, O, (3, 3), 1, 1, bias=kargs['bias'])
                if not kargs['bias']:
                    raise NotImplementedError('bias=False: main_train.py:174 hard-codes bias=True')
            else:
                self.rbr_3x3_branch = nn.Conv2d(C, O, (3, 3), 1, 1)
                self.rbr_3x1_branch = nn.Conv2d(C, O, (3, 1), 1, (1, 0))
                self.rbr_1x3_branch = nn.Conv2d(C, O, (1, 3), 1, (0, 1))
                self.rbr_1x1_3x3_1x1_branch_1x1_1 = nn.Conv2d(C, 2 * C, (1, 1), 1, 0, bias=False)
                self.rbr_1x1_3x3_1x1_branch_3x3 = nn.Conv2d(2 * C, O, (3, 3), 1, 1, bias=False)
                self.rbr_1x1_3x3_1x1_branch_1x1_2 = nn.Conv2d(O, O, (1, 1), 1, 0, bias=False)

    # model.py:450-478
    def get_equivalent_kernel_bias(self):
        return ops.ErbMergeFn.apply(
            self.rbr_3x3_branch.weight, self.rbr_3x3_branch.bias,
            self.rbr_3x1_branch.weight, self.rbr_3x1_branch.bias,
            self.rbr_1x3_branch.weight, self.rbr_1x3_branch.bias,
            self.rbr_1x1_3x3_1x1_branch_1x1_1.weight, self.rbr_1x1_3x3_1x1_branch_3x3.weight,
            self.rbr_1x1_3x3_1x1_branch_1x1_2.weight)

    # model.py:395-448
    def switch_to_deploy(self):
        if getattr(self, 'deploy', False) or not hasattr(self, 'rbr_3x3_branch'):
            if hasattr(self, 'rbr_reparam'):
                self.deploy = True
            return
        with torch.no_grad():
            kernel, bias = self.get_equivalent_kernel_bias()
        if not hasattr(self, 'rbr_reparam'):
            self.rbr_reparam = nn.Conv2d(self.ngf, self.out_channels, (3, 3), 1, 1, bias=True)
        self.rbr_reparam.weight.data = kernel.detach()
        self.rbr_reparam.bias.data = bias.detach()
        for name in ['rbr_3x3_branch', 'rbr_3x1_branch', 'rbr_1x3_branch', 'rbr_1x1_3x3_1x1_branch_1x1_1',
                     'rbr_1x1_3x3_1x1_branch_3x3', 'rbr_1x1_3x3_1x1_branch_1x1_2', 'branch']:
            if hasattr(self, name):
                self.__delattr__(name)
        self.deploy = True

    def switch_to_deploy_structure(self):
        """Module surgery of switch_to_deploy without computing the merge (the deploy checkpoint supplies the
        merged kernel): creates rbr_reparam, drops the training branches.  No GPU needed."""
        if getattr(self, 'deploy', False):
            return
        if not hasattr(self, 'rbr_reparam'):
            self.rbr_reparam = nn.Conv2d(self.ngf, self.out_channels, (3, 3), 1, 1, bias=True)
        for name in ['rbr_3x3_branch', 'rbr_3x1_branch', 'rbr_1x3_branch', 'rbr_1x1_3x3_1x1_branch_1x1_1',
                     'rbr_1x1_3x3_1x1_branch_3x3', 'rbr_1x1_3x3_1x1_branch_1x1_2', 'branch']:
            if hasattr(self, name):
                self.__delattr__(name)
        self.deploy = True

    # model.py:518-567
    def forward(self, x):
        if self.deploy:
            w, b = self.rbr_reparam.weight, self.rbr_reparam.bias
        elif self.branch_type == 'NeRV_vanilla':
            w, b = self.branch.weight, self.branch.bias
        else:
            w, b = self.get_equivalent_kernel_bias()
        return ops.ConvPsSiluFn.apply(x, w, b, self.stride)


class Generator(nn.Module):
    """model.py:571-625 (num_blocks = 1 per stage, as every BASELINE config uses)."""

    def __init__(self, **kargs):
        super().__init__()
        stem_dim, stem_num = [int(x) for x in kargs['stem_dim_num'].split('_')]
        self.fc_h, self.fc_w, self.fc_dim = [int(x) for x in kargs['fc_hw_dim'].split('_')]
        _only(kargs.get('act', 'swish'), ('swish',), 'act')
        if stem_num != 1:
            raise NotImplementedError('stem_dim_num with more than one hidden layer is outside the hot path')
        if kargs.get('num_blocks', 1) != 1:
            raise NotImplementedError('num_blocks != 1 is outside the hot path')
        self.embed_length = kargs['embed_length']
        # nn.Sequential(Linear, act, Linear, act) -> keys stem.0.*, stem.2.* (model.py:183-188)
        self.stem = nn.Sequential(nn.Linear(kargs['embed_length'], stem_dim), nn.SiLU(inplace=True),
                                  nn.Linear(stem_dim, self.fc_h * self.fc_w * self.fc_dim), nn.SiLU(inplace=True))
        self.layers, self.head_layers = [nn.ModuleList() for _ in range(2)]
        ngf = self.fc_dim
        self.stride_list = list(kargs['stride_list'])
        for i, stride in enumerate(self.stride_list):
            if i == 0:
                new_ngf = int(ngf * kargs['expansion'])
            else:
                new_ngf = max(ngf // (1 if stride == 1 else kargs['reduction']), kargs['lower_width'])
            self.layers.append(NeRVBlock(ngf=ngf, new_ngf=new_ngf, stride=stride, bias=kargs['bias'], norm=kargs['norm'],
                                         act=kargs['act'], deploy=kargs['deploy'], conv_type=kargs.get('conv_type', 'conv'),
                                         branch_type=kargs['branch_type']))
            ngf = new_ngf
            head_layer = None
            if kargs['sin_res']:
                if i == len(self.stride_list) - 1:
                    head_layer = nn.Conv2d(ngf, 3, 1, 1, bias=kargs['bias'])
            else:
                raise NotImplementedError('multi-resolution heads (no --single_res) are outside the hot path')
            self.head_layers.append(head_layer)
        self.sigmoid = kargs['sigmoid']

    def forward(self, input):
        s0, s2 = self.stem[0], self.stem[2]
        output = ops.StemFn.apply(input, s0.weight, s0.bias, s2.weight, s2.bias)
        output = output.view(output.size(0), self.fc_dim, self.fc_h, self.fc_w)
        out_list = []
        for layer, head_layer in zip(self.layers, self.head_layers):
            output = layer(output)
            if head_layer is not None:
                out_list.append(ops.HeadFn.apply(output, head_layer.weight, head_layer.bias, self.sigmoid))
        return out_list
