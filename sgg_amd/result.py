"""`Result` container with the reference's semantics (lib/pytorch_misc.py:682-708): keyword fields, the ones
left as None are removed so that callers test with hasattr() (sgg_models/rel_model_stanford.py:141)."""


class Result(object):
    _FIELDS = ('od_obj_dists', 'rm_obj_dists', 'obj_scores', 'obj_preds', 'obj_fmap', 'od_box_deltas',
               'rm_box_deltas', 'od_box_targets', 'rm_box_targets', 'od_box_priors', 'rm_box_priors',
               'boxes_assigned', 'boxes_all', 'od_obj_labels', 'rm_obj_labels', 'rpn_scores', 'rpn_box_deltas',
               'rel_labels', 'rel_labels_all', 'im_inds', 'fmap', 'rel_dists', 'rel_inds', 'rel_rep')

    def __init__(self, **kwargs):
        for k in kwargs:
            if k not in self._FIELDS:
                raise TypeError("__init__() got an unexpected keyword argument '%s'" % k)
        for k in self._FIELDS:
            v = kwargs.get(k)
            if v is not None:
                self.__dict__[k] = v

    def is_none(self):
        return all([v is None for k, v in self.__dict__.items() if k != 'self'])

    def __getitem__(self, index):
        d = self.__dict__
        values = [d[k] for k in sorted(list(d.keys()))]
        return values[index]
