(* GPHIPSampler.wl -- nestedSamplingHIP: the library's native batched nested-sampling driver behind the reference's own result
   shape.  Loaded by GPHIP.wl (inside GPHIP`Private`: it uses that package's handle bookkeeping).  Like GPHIP.wl it cannot be
   executed in the build containers; the numerics it calls are tested through the shim (tests/test_gpu_wl_shim.py), the pool
   drawing and the table look-up in C (tests/test_host_logic.py). *)

gpNested := gpNested = load["gphip_wl_nested_sampling", {Integer, m2, iv, v1, any}, {Real, 2}];
gpNestedTab := gpNestedTab = load["gphip_wl_nested_sampling_tab", {Integer, m2, m2, v1, any}, {Real, 2}];
gpNestedCb := gpNestedCb = load["gphip_wl_nested_sampling_cb", {Integer, m2, v1, m2}, {Real, 2}];

(* ---- the native batched sampler (gphip_nested_sampling): nestedSamplingInternal's job (BayesianStatistics.wl:859-1040)
   done inside the library, `Walkers` constrained-prior chains in lock step = one batched likelihood call per Metropolis
   step instead of "MonteCarloSteps" sequential calls of the closure.  The rows come back in generation order; they are
   wrapped into the reference's "Samples" association (:903-913, :1005-1013) and post-processed by the reference's OWN
   evidenceSampling (:1158-1291), so the returned object has exactly the keys nestedSampling gives. ---- *)
Options[nestedSamplingHIP] = Join[Options[nestedSampling], {"Walkers" -> 32, "Seed" -> 0, "PriorKinds" -> Automatic}];

uniformPriorQ[prior_, p_] := MatchQ[prior, "Uniform" | _UniformDistribution |
	ProductDistribution[(_UniformDistribution | {_UniformDistribution, _Integer})..]];

(* Any SEPARABLE prior (a ProductDistribution of univariate distributions -- or one univariate distribution for one
   parameter): its "LogPriorPDFFunction" (BayesianStatistics.wl:256-274) is the sum of the factors' log densities, which travel
   to the native driver as tables on a uniform grid over each parameter's {min, max}; zeros of a density become -1.*^300 (read
   as -Infinity by the shim); the starting pool is drawn from the tables by the library (generateStartingPoints, BS:1046-1068).
   $Failed for anything else.  A JOINT prior travels as a LibraryLink callback instead: the object's compiled
   "LogPriorPDFFunction" (BS:412-427) is connected to the library's "gphip_logprior" manager and evaluated by the native driver
   once per proposed point; its starting pool comes from the reference's own generateStartingPoints (BS:1055-1068). *)
connectPrior[f_CompiledFunction] := (gpNestedCb; TrueQ @ Quiet @ ConnectLibraryCallbackFunction["gphip_logprior", f]);
connectPrior[_] := False;
priorFactors[ProductDistribution[d__], p_] := With[{f = Flatten[Replace[{d}, {dist_, n_Integer} :> ConstantArray[dist, n], {1}]]},
	If[Length[f] === p && AllTrue[f, UnivariateDistributionQ], f, $Failed]];
priorFactors[d_?UnivariateDistributionQ, 1] := {d};
priorFactors[__] := $Failed;
priorTables[factors_List, params_, m_Integer] := MapThread[
	Function[{dist, spec},
		Clip[Replace[N @ Log @ PDF[dist, N @ Subdivide[spec[[2]], spec[[3]], m - 1]], Except[_Real] -> -1.*^300, {1}], {-1.*^300, 1.*^300}]],
	{factors, params}];
nestedSamplingHIP[inferenceObject[assoc_?AssociationQ], opts : OptionsPattern[]] /;
	KeyExistsQ[Lookup[assoc, "GaussianProcessData", <||>], "HIPHandle"] := Module[{
	h = assoc["GaussianProcessData", "HIPHandle"],
	params = assoc["Parameters"], p, pool, kinds, factors, joint = False, nsOpts, start, rows, samples, result,
	pwFlags = Lookup[assoc["GaussianProcessData"], "PointwiseFunctions", {False, False}]
},
	p = Length[params];
	kinds = Replace[OptionValue["PriorKinds"], Automatic :> If[uniformPriorQ[assoc["PriorDistribution"], p], ConstantArray[0, p], $Failed]];
	factors = If[kinds === $Failed, priorFactors[assoc["PriorDistribution"], p], $Failed];
	start = Replace[OptionValue["StartingPoints"], Except[_?(MatrixQ[#, NumericQ]&)] :> Lookup[assoc, "StartingPoints", {}]];
	If[ kinds === $Failed && factors === $Failed && !Or @@ pwFlags,
		If[ !MatrixQ[start], start = generateStartingPoints[assoc, OptionValue["SamplePoolSize"]]];
		joint = MatrixQ[start, NumericQ] && connectPrior[assoc["LogPriorPDFFunction"]]
	];
	(* point-dependent nugget / mean functions live in this kernel process, and so does a prior that is not a compiled
	   function of the parameter vector: those runs go through the reference's own driver with the GPU closure *)
	If[ (kinds === $Failed && factors === $Failed && !joint) || Or @@ pwFlags,
		Return @ nestedSampling[inferenceObject[assoc], Sequence @@ FilterRules[{opts}, Options[nestedSampling]]]
	];
	pool = If[MatrixQ[start], Length[start], OptionValue["SamplePoolSize"]];
	touch[h];
	nsOpts = N @ {pool, OptionValue["MaxIterations"], OptionValue["MinIterations"], OptionValue["MonteCarloSteps"], OptionValue["Walkers"],
		OptionValue["TerminationFraction"], Sequence @@ OptionValue["MinMaxAcceptanceRate"], OptionValue["Seed"]};
	rows = Which[
		joint, gpNestedCb[h, N @ params[[All, {2, 3}]], nsOpts, N @ start],
		kinds =!= $Failed, gpNested[h, N @ params[[All, {2, 3}]], kinds, nsOpts, If[MatrixQ[start], N @ start, {}]],
		(* separable prior: tabulated factors; without starting points the library draws the pool from the tables itself *)
		True, gpNestedTab[h, N @ params[[All, {2, 3}]], priorTables[factors, params, 2049], nsOpts, If[MatrixQ[start], N @ start, {}]]
	];
	If[ !MatrixQ[rows], Return["Bad likelihood function"]];     (* BayesianStatistics.wl:917-921 *)
	samples = Association @ MapIndexed[
		Function[{row, index}, First[index] -> <|"Point" -> row[[;; p]], "LogLikelihood" -> row[[p + 1]], "LogPriorPDF" -> row[[p + 2]],
			"AcceptanceRate" -> If[First[index] <= pool, Missing["InitialSample"], row[[p + 3]]]|>],
		rows];
	result = evidenceSampling[
		<|"Samples" -> samples, "SamplePoolSize" -> pool, "GeneratedNestedSamples" -> Length[rows] - pool, "TotalSamples" -> Length[rows],
			"ParameterRanges" -> CoordinateBounds[rows[[All, ;; p]]]|>,
		params[[All, 1]], Sequence @@ FilterRules[{opts}, Options[evidenceSampling]]];
	If[ TrueQ @ AssociationQ[result],
		inferenceObject[Join[assoc, <|"StartingPoints" -> rows[[;; pool, ;; p]]|>, result]],
		result
	]
];
nestedSamplingHIP[obj_, opts___] := nestedSampling[obj, Sequence @@ FilterRules[{opts}, Options[nestedSampling]]];

