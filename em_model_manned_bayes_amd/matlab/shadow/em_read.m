function parms = em_read(parameters_filename, varargin)
% EM_READ  Same signature and struct as the reference's em_read (em_read.m:1,41-42,47-141), parsed by libemgpu's loader.
% Put this directory AHEAD of the reference's code/matlab on the MATLAB path to shadow it.  UNTESTED (no MATLAB here).
    p = inputParser;
    addRequired(p, 'parameters_filename');
    addParameter(p, 'isOverwriteZeroBoundaries', false, @islogical);
    addParameter(p, 'idxZeroBoundaries', [1 2 3], @isnumeric);
    parse(p, parameters_filename, varargin{:});
    parms = emgpu_mex('em_read', char(parameters_filename), p.Results.idxZeroBoundaries, p.Results.isOverwriteZeroBoundaries);
end
